/*
 * lerc_amd_device.h -- the part of liblerc_amd.so's C ABI that has NO counterpart in Esri/lerc: the codec of
 * lerc_amd.h on DEVICE pointers (synchronous, stream-asynchronous, batched tile mosaics) plus diagnostics.
 * The stock twelve lerc_* entry points are declared in lerc_amd.h (which includes this file).
 */
#ifndef LERC_AMD_DEVICE_H
#define LERC_AMD_DEVICE_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef LERC_AMD_API
#define LERC_AMD_API __attribute__((visibility("default")))
typedef unsigned int lerc_status;
#endif

typedef struct lerc_amd_context lerc_amd_context;

/* hipStream: a hipStream_t cast to void* on which all work is enqueued; NULL is the HIP default
 * (NULL) stream, with its usual ordering against other blocking streams.  A context owns its scratch HBM; use one context per host thread. */
LERC_AMD_API lerc_amd_context* lerc_amd_create(void* hipStream);
LERC_AMD_API void lerc_amd_destroy(lerc_amd_context* ctx);
LERC_AMD_API void lerc_amd_set_stream(lerc_amd_context* ctx, void* hipStream);
LERC_AMD_API const char* lerc_amd_last_error(lerc_amd_context* ctx);

/* Same contracts as lerc_encode / lerc_decode, but pData, pValidBytes, pOutBuffer / pLercBlob are
 * DEVICE pointers.  dOutBuffer == NULL turns lerc_amd_encode_device into the exact size query.
 * The calls return after the stream has been synchronised (the blob size is a host-visible result). */
LERC_AMD_API lerc_status lerc_amd_encode_device(lerc_amd_context* ctx, const void* dData, unsigned int dataType,
    int nDepth, int nCols, int nRows, int nBands, int nMasks, const unsigned char* dValidBytes, double maxZErr,
    unsigned char* dOutBuffer, unsigned int outBufferSize, unsigned int* nBytesWritten);
LERC_AMD_API lerc_status lerc_amd_decode_device(lerc_amd_context* ctx, const unsigned char* dLercBlob,
    unsigned int blobSize, int nMasks, unsigned char* dValidBytes, int nDepth, int nCols, int nRows, int nBands,
    unsigned int dataType, void* dData);

/* The same two calls without the wait: the operation is enqueued on the context's stream and the call returns a ticket.
 * Operations of one context run in the order they were enqueued, so a decode may be enqueued right behind the encode
 * that writes its blob: pass the CAPACITY of the blob buffer as blobSizeBound, the true size is read from the header on
 * the device.  lerc_amd_finish(ctx, ticket, &n) waits for the stream, returns that operation's lerc_status (and, for an
 * encode, the bytes written / needed) and forgets it; ticket 0 waits for everything and drops all results.  Requests
 * the streaming kernels do not take blind (masks, several bands, nDepth > 1, ...) are carried out inside the _async call
 * itself, in order.  If the device hands an operation back to the general path (a constant raster, a damaged blob,
 * ...), lerc_amd_finish repeats it and everything enqueued behind it synchronously -- results are the same as with
 * the synchronous calls, only later.  Buffers must stay untouched until the operation has been finished.  At most 31
 * unfinished results are kept; older ones are completed and dropped. */
LERC_AMD_API lerc_status lerc_amd_encode_device_async(lerc_amd_context* ctx, const void* dData, unsigned int dataType,
    int nDepth, int nCols, int nRows, int nBands, int nMasks, const unsigned char* dValidBytes, double maxZErr,
    unsigned char* dOutBuffer, unsigned int outBufferSize, unsigned int* ticket);
LERC_AMD_API lerc_status lerc_amd_decode_device_async(lerc_amd_context* ctx, const unsigned char* dLercBlob,
    unsigned int blobSizeBound, int nMasks, unsigned char* dValidBytes, int nDepth, int nCols, int nRows, int nBands,
    unsigned int dataType, void* dData, unsigned int* ticket);
LERC_AMD_API lerc_status lerc_amd_finish(lerc_amd_context* ctx, unsigned int ticket, unsigned int* nBytes);

/* Tile mosaics: nTiles rasters of one shape, contiguous on the device ([nTiles][nRows][nCols], 1 band, nDepth 1,
 * no masks), in ONE call (tiles are independent blobs; ranks of a multi-GPU job take tile ranges).
 * Tile t becomes exactly the blob lerc_encode() would make of it, at dArena + offsets[t] (16-byte aligned), sizes[t]
 * bytes long; offsets / sizes / arenaUsed are HOST arrays the caller provides.  BufferTooSmall(3) if the arena is
 * too small (lerc_computeCompressedSize bounds a tile; nRows * nCols * sizeof(T) + 128 per tile always suffices).
 * Decoding takes the same description back.  Blobs that need the general kernels are handled inside, one by one. */
LERC_AMD_API lerc_status lerc_amd_encode_tiles_device(lerc_amd_context* ctx, const void* dTiles, unsigned int dataType, int nCols,
    int nRows, int nTiles, double maxZErr, unsigned char* dArena, unsigned long long arenaCapacity, unsigned long long* offsets,
    unsigned int* sizes, unsigned long long* arenaUsed);
LERC_AMD_API lerc_status lerc_amd_decode_tiles_device(lerc_amd_context* ctx, const unsigned char* dArena,
    const unsigned long long* offsets, const unsigned int* sizes, int nTiles, int nCols, int nRows, unsigned int dataType, void* dTiles);

/* The same with a slot per tile, the way a caller of lerc_encode() hands every tile a buffer of its own: tile t's blob goes to
 * dSlots + t * slotBytes (slotBytes: a multiple of 16, the capacity of every slot), sizes[t] bytes long -- nothing is moved
 * behind the encode kernel (the packed form costs one more pass over the blobs).  BufferTooSmall(3) if a tile's blob does
 * not fit its slot. */
LERC_AMD_API lerc_status lerc_amd_encode_tiles_device_slots(lerc_amd_context* ctx, const void* dTiles, unsigned int dataType, int nCols,
    int nRows, int nTiles, double maxZErr, unsigned char* dSlots, unsigned long long slotBytes, unsigned int* sizes);
LERC_AMD_API lerc_status lerc_amd_decode_tiles_device_slots(lerc_amd_context* ctx, const unsigned char* dSlots, unsigned long long slotBytes,
    const unsigned int* sizes, int nTiles, int nCols, int nRows, unsigned int dataType, void* dTiles);

/* Per-kernel timing with HIP events on the context's stream (used by bench.py for the roofline of the
 * dominant kernel).  lerc_amd_profile_read writes lines "kernel_group total_ms launches" into buf. */
LERC_AMD_API void lerc_amd_profile_enable(lerc_amd_context* ctx, int on);
LERC_AMD_API int lerc_amd_profile_read(lerc_amd_context* ctx, char* buf, int cap, int reset);

/* Which kernels served the successful calls of a context so far: out[0] encodes by the streaming kernels, out[1]
 * encodes by the general kernels, out[2] / out[3] the same for decodes.  ctx == NULL: the calling thread's context
 * behind lerc_encode / lerc_decode.  Diagnostics only (tests assert that the streaming path really ran). */
LERC_AMD_API void lerc_amd_path_counters(lerc_amd_context* ctx, unsigned long long out[4]);
/* Which of the streaming decoders served the bands / tiles counted in out[2] above: out[3] the scanning decoder (one launch, no
 * walks), out[2] the walking one-launch decoder, out[1] discovery + decode in two launches; out[0] counts bands WITH a mask whose blocks the scanning decoder's first half found
 * (the general kernels decode their pixels).  Same ctx convention. */
LERC_AMD_API void lerc_amd_decode_forms(lerc_amd_context* ctx, unsigned long long out[4]);
/* The mosaic job's ONE exchange step (the reference has no counterpart: whoever tiles a mosaic moves the blobs
 * himself): the gather of the ranks' compressed blobs on the rank that writes the container, over RCCL -- xGMI inside a node.
 *   ncclComm      an ncclComm_t of the caller's (ncclCommInitRank; librccl is opened with dlopen when this is first called, inside a
 *                 PyTorch process that is the RCCL torch.distributed uses); rank and size are the communicator's
 *   dMessage      this rank's message, nBytes of device memory: its arena as lerc_amd_encode_tiles_device left it -- with the per-tile
 *                 (offset, size) table in front, if the caller laid it out so: one message a rank
 *   dRootBuffer   on the root: where the messages land, rank r's at hOffsets[r] (16-byte aligned, back to back); ignored elsewhere
 *   hLengths      out, host, one word a rank: every rank's message length (all ranks get them); may be NULL
 *   hOffsets      out, host, ranks + 1 words; may be NULL
 *   stream        the HIP stream everything is enqueued on: one ncclAllGather of the lengths, then a sender's ncclSend at once (it needs
 *                 nobody's length but its own), on the root one group of ncclRecv behind the only host wait there is (the lengths, 8 bytes
 *                 a rank through pinned memory) and a copy of its own message.  The call returns when the transfers are POSTED; wait
 *                 for the stream before the bytes are read.  Order it behind the encodes' stream with an event.
 * Returns 0, 2 (WrongParam), 3 (BufferTooSmall: rootCapacity), 1 (Failed: lerc_amd_last_error says which RCCL call). */
LERC_AMD_API unsigned int lerc_amd_gather_blobs(lerc_amd_context* ctx, void* ncclComm, int root, const void* dMessage, unsigned long long nBytes,
                                                void* dRootBuffer, unsigned long long rootCapacity, unsigned long long* hLengths,
                                                unsigned long long* hOffsets, void* stream);
/* Attempts that were thrown away on the way down the tiers -- each is a launch (or several) whose result nobody used: out[0] the decode
 * kernels refused the block offsets the scan had proposed for a band with a mask (a raw block's length is a guess there; the general
 * discovery then takes the band), out[1] that scan handed a masked band on by itself, out[2] a streaming decode tier (scan, walk, two
 * launches) handed a band to the next one; out[3] unused.  Same ctx convention. */
LERC_AMD_API void lerc_amd_decode_refusals(lerc_amd_context* ctx, unsigned long long out[4]);
/* why the last call that left the streaming kernels did so ("" if none did); same ctx convention */
LERC_AMD_API const char* lerc_amd_last_note(lerc_amd_context* ctx);

/* The run-length coding of a validity BIT mask (the mask section of a Lerc2 blob, RLE.cpp:123-254: nBytes = (nPix + 7) / 8
 * bytes, most significant bit first) on the device, as the masked encode uses it: dBits and dOut device pointers (dBits 16-byte
 * aligned), *size = bytes written incl. the end marker.  Returns 0, or 3 (BufferTooSmall) if the stream does not fit cap. */
LERC_AMD_API unsigned int lerc_amd_mask_rle_device(lerc_amd_context* ctx, const unsigned char* dBits, unsigned int nBytes,
                                                   unsigned char* dOut, unsigned int cap, unsigned int* size);

/* The other way, as the masked decode uses it: nBytes mask bytes out of a stream of rleBytes bytes (RLE.cpp:259-330); what the
 * stream does not hold stays zero.  Returns 0, or 1 (Failed) for a damaged stream (no end marker, a segment that runs over
 * either end). */
LERC_AMD_API unsigned int lerc_amd_mask_rle_decode_device(lerc_amd_context* ctx, const unsigned char* dRle, unsigned int rleBytes,
                                                          unsigned char* dBits, unsigned int nBytes);

/* library / build identification: "lerc_amd <version> gfx950 hip" (or "... hipsim" for the CPU test build) */
LERC_AMD_API const char* lerc_amd_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
