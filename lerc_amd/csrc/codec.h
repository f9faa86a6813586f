// codec.h -- host side of the MI355X LERC path: per-call context, header / mask / mode-decision logic
// and the kernel pipelines behind lerc_encode() / lerc_decode().
//
// Reference counterparts: Lerc::EncodeInternal (Lerc.cpp:628-789), Lerc::DecodeTempl (:397-521),
// Lerc2::ComputeNumBytesNeededToWrite (Lerc2.cpp:179-381), Lerc2::Encode (:396-480), Lerc2::Decode
// (:577-694).  Pixel-touching work is never done here: it is enqueued as HIP kernels (kernels.h).
#pragma once
#include "kernels.h"

#include <string>
#include <functional>
#include <vector>

namespace lerc {

// Lerc2::HeaderInfo (Lerc2.h:102-131)
struct Header
{
  int version = kCodecVersion;
  u32 checksum = 0;
  int nRows = 0, nCols = 0, nDepth = 1, numValid = 0, mbSize = 8, blobSize = 0, dt = DT_Undefined, nBlobsMore = 0;
  u8 passNoData = 0, isInt = 0, rsv3 = 0, rsv4 = 0;
  double maxZErr = 0, zMin = 0, zMax = 0, noDataVal = 0, noDataValOrig = 0;
  bool tryHuffmanInt() const { return version >= 2 && (dt == DT_Byte || dt == DT_Char) && maxZErr == 0.5; }
  bool tryHuffmanFlt() const { return version >= 6 && (dt == DT_Float || dt == DT_Double) && maxZErr == 0; }
};
u32 headerBytes(int version);
void writeHeader(u8* dst, const Header& h);
bool readHeader(const u8* src, size_t n, Header& h, size_t& used);

// RLE of the validity bit mask (host; the mask is << 1 % of the bytes and inherently sequential)
void rleEncode(const u8* src, size_t n, std::vector<u8>& out);
bool rleEncodeWhenReady(const u8* b, size_t n, const std::function<bool()>& ready, std::vector<u8>& out);    // (pieces by several threads, started before the bytes are there)
bool rleDecode(const u8* src, size_t n, u8* dst, size_t dstSize, size_t* written = nullptr);    // (what the stream does not fill stays as it was)

// ---- growable device workspace + stream, one per host thread (C API) or per handle (device API)
class Context
{
public:
  Context();
  ~Context();
  bool ok() const { return m_ok; }
  hipStream_t stream() const { return m_stream; }
  // a caller-provided stream; nullptr means the HIP default (NULL) stream, exactly like a HIP API call would
  void setStream(hipStream_t s) { m_userStream = s; m_userSet = true; }
  hipStream_t activeStream() const { return m_userSet ? m_userStream : m_stream; }

  // bump allocation out of one device slab; reserve() may re-allocate (invalidates earlier pointers)
  bool reserve(size_t bytes);
  void poisonScratch(size_t bytes);          // test aid, see codec_common.cpp
  void reset() { m_used = 0; }
  size_t used() const { return m_used; }
  void rewind(size_t mark) { m_used = mark; }    // gives back everything allocated since used() returned mark
  void* alloc(size_t bytes, size_t align = 256);
  template<class T> T* allocT(size_t n) { return (T*)alloc(n * sizeof(T)); }

  // small pinned host mirror for results read back after a sync
  void* pinned(size_t bytes);
  // pinned result slots of operations that are enqueued but not waited for yet (asynchronous device API)
  static const int kAsyncSlots = 64;
  static const size_t kAsyncSlotBytes = 512;
  u8* asyncSlot(unsigned ticket);
  // a second pinned area (the validity bits of a band on their way to the host while kernels run) and its event
  void* pinnedAux(size_t bytes);
  hipEvent_t auxEvent();
  // a stream beside the active one for that copy (it branches off behind what is enqueued so far: forkSide()); nullptr: none to be had
  hipStream_t forkSide();
  // more work has gone onto that stream since the last sync(): the next sync() / reset() waits for it again
  void sideInUse() { if (m_sideStream) m_sideUsed = true; }
  bool sync();                               // wait for the active stream (polls first: see codec_common.cpp)
  // Device memory that keeps its contents from call to call (the two-launch encoder's arrival counters and cells,
  // tile_fast.h): zero when handed out for the first time and whenever it had to grow.  Two areas: [0] counters, which the
  // kernels leave zero, and [1] cells tagged with the call's epoch, which they leave as they are.
  u8* persistentState(int area, size_t bytes);
  // after a kernel has reported a hand-off it gave up on ("stuck"): late workgroups may have left residue in the counters,
  // so both areas are wiped (after waiting for the stream) before the next call uses them
  void wipePersistentState();
  // a value no earlier call of this context has used and that no fill pattern looks like: kernels raise flags by
  // writing it into cells that are never cleared (tile_fast.h)
  u32 nextEpoch() { m_epoch += 0x9E3779B9u; if ((m_epoch & 0xFFFFu) == (m_epoch >> 16) || m_epoch == 0u) m_epoch += 0x9E3779B9u; return m_epoch; }

  std::string lastError;
  std::string lastNote;      // diagnostics that are not errors (why a call left the streaming path)

  // which kernels served the calls so far: [0] encode streaming, [1] encode general, [2] decode streaming, [3] decode general
  unsigned long long pathCount[4] = { 0, 0, 0, 0 };
  bool lastDecodeStreamed = false;
  unsigned long long formCount[4] = { 0, 0, 0, 0 };    // bands / tiles decoded by streaming form 1, 2, 3 (lerc_amd_decode_forms)
  unsigned long long refusalCount[4] = { 0, 0, 0, 0 };    // attempts thrown away (lerc_amd_decode_refusals): [0] the decode kernels refused the masked scan's block offsets, [1] the masked scan handed a band on, [2] a streaming decode tier handed a band on, [3] unused
  int lastStreamForm = 0;              // the form decodeEnqueueStreaming last enqueued (DecodeRequest::maxForm)
  struct { int dt = -1, nRows = 0, nCols = 0; u32 end = 0; } scanHint;    // size of the last band the streaming kernels decoded, and its shape (see launchFastBands)
  int lastStreamShape[3] = { -1, 0, 0 };                                 // dt, nRows, nCols of the band decodeEnqueueStreaming / decodeImpl last enqueued
  u32 blindSkip = 0;                   // decodes of blindShape that go to the header-reading path at once (the last blind attempt was refused by the header)
  int blindShape[3] = { -1, 0, 0 };
  bool scanOffsetsBan = false;         // this call: a masked band's scan for block offsets has failed, the general discovery takes the bands
  u32 scanLateSpan = 64;               // how many decodes the next wrong early count keeps the context on late counts (grows: 64, 256, ... 4096)
  int scanLateRows = 0, scanLateCols = 0;    // ... and the shape of the band whose early count was wrong: the window is for bands of that shape
  u32 scanLate = 0;                    // decodes whose scanning decoder counts late: an early count has just turned out wrong (a stream with blocks the scan does not see: the mending found them)
  u32 lastScanGridBytes = 0;           // blob bytes the scanning decoder's last launch held pieces for (launchFastBands)
  int scanSkipRows = 0, scanSkipCols = 0;    // ... of that shape
  u32 scanSkip = 0;                    // decodes that keep off the scanning decoder: it has just handed a band on (a stream with blocks it cannot see)

  // optional per-kernel timing with HIP events on the active stream (bench.py: roofline of the dominant kernel)
  void profEnable(bool on) { m_prof = on; }
  bool profOn() const { return m_prof; }
  void profBegin(const char* name);
  void profEnd();
  void profCollect();                        // call after a stream sync
  std::string profReport(bool reset);        // "name total_ms launches" per line

private:
  struct ProfEntry { const char* name; hipEvent_t a, b; bool ownsA, ownsB; };
  bool m_lastEndFresh = false;               // nothing was enqueued since the last profEnd()
  struct ProfAcc { std::string name; double ms; int n; };
  bool m_prof = false;
  std::vector<ProfEntry> m_pending;
  std::vector<hipEvent_t> m_eventPool;
  std::vector<ProfAcc> m_acc;
  hipEvent_t profEvent();

  u8* m_asyncPinned = nullptr;
  u8* m_state[2] = { nullptr, nullptr };
  size_t m_stateCap[2] = { 0, 0 };
  unsigned long long m_stateCalls = 0;
  bool m_ok = false;
  u32 m_epoch = 0x1234567u;
  hipStream_t m_stream = nullptr, m_userStream = nullptr;
  bool m_userSet = false;
  u8* m_slab = nullptr;
  size_t m_cap = 0, m_used = 0;
  void* m_pinned = nullptr;
  size_t m_pinnedCap = 0;
  void* m_pinnedAux = nullptr;
  size_t m_pinnedAuxCap = 0;
  hipEvent_t m_auxEvent = nullptr, m_forkEvent = nullptr;
  hipStream_t m_sideStream = nullptr;
  bool m_sideUsed = false;
};

// RAII bracket around one kernel launch (or a short group of launches)
struct ProfScope
{
  Context& c;
  ProfScope(Context& ctx, const char* name) : c(ctx) { if (c.profOn()) c.profBegin(name); }
  ~ProfScope() { if (c.profOn()) c.profEnd(); }
};

// the mosaic job's exchange step over RCCL (gather_rccl.cpp)
u32 gatherBlobsRccl(void* comm, int root, const void* dMessage, u64 nBytes, void* dRootBuffer, u64 rootCapacity, u64* hLengths, u64* hOffsets,
                    hipStream_t st, std::string& err);

// ---- whole-call entry points on DEVICE-resident pixel data -------------------------------------
struct EncodeRequest
{
  const void* dData = nullptr;        // device: [nBands][nRows][nCols][nDepth]
  const u8* dValidBytes = nullptr;    // device byte masks (nMasks of them) or nullptr
  int dt = DT_Undefined, nDepth = 1, nCols = 0, nRows = 0, nBands = 1, nMasks = 0;
  double maxZErr = 0;
  u8* dOut = nullptr;                 // device output buffer (nullptr: size only)
  u32 outCapacity = 0;
  const u8* hUsesNoData = nullptr;    // host [nBands] or nullptr: band carries a noData value (lerc_encode_4D)
  const double* hNoDataValues = nullptr;
  int version = kCodecVersion;        // codec version of the blobs to write: 3..6 (lerc_encodeForVersion)
};
// returns an ErrCode; numBytesNeeded is always the exact blob size on kOk
u32 encodeDevice(Context& ctx, const EncodeRequest& rq, u32& numBytesNeeded, u32& numBytesWritten);

// nTiles rasters of one shape, contiguous on the device, each to become (or coming from) its own blob in an arena
struct TilesEncodeRequest
{
  const void* dData = nullptr;        // device: [nTiles][nRows][nCols]
  int dt = 0, nCols = 0, nRows = 0, nTiles = 0;
  double maxZErr = 0;
  u8* dArena = nullptr;               // device
  u64 arenaCapacity = 0;
  u64* hOffsets = nullptr;            // host [nTiles]: where tile t's blob starts in the arena (16-byte aligned)
  u32* hSizes = nullptr;              // host [nTiles]
  u64 slotBytes = 0;                  // != 0: tile t's blob goes to dArena + t * slotBytes (a multiple of 16), nothing is moved afterwards
};
struct TilesDecodeRequest
{
  const u8* dArena = nullptr;         // device
  const u64* hOffsets = nullptr;      // host [nTiles]
  const u32* hSizes = nullptr;        // host [nTiles]
  int dt = 0, nCols = 0, nRows = 0, nTiles = 0;
  void* dOut = nullptr;               // device: [nTiles][nRows][nCols]
};

struct DecodeRequest
{
  const u8* hBlob = nullptr;          // host copy of the blob (may be nullptr when only dBlob is known)
  const u8* dBlob = nullptr;          // device copy of the blob (nullptr: staged from hBlob)
  u32 blobSize = 0;
  int dt = DT_Undefined, nDepth = 1, nCols = 0, nRows = 0, nBands = 1, nMasks = 0;
  void* dOut = nullptr;               // device: decoded pixels
  u8* dValidBytes = nullptr;          // device: nMasks byte masks, or nullptr
  bool noStreaming = false;            // go straight to the general kernels (a batch has already tried the streaming ones)
  int maxForm = 4;                     // the first streaming form to try: 4 the scanning decoder with early counts, 3 the scanning decoder, 2 the walking one-launch decoder, 1 discovery + decode in
                                       // two launches (a form that has just refused this blob hands it on with its own number less one)
  u8* hUsesNoData = nullptr;           // host [nBands] out (lerc_decode_4D), or nullptr
  double* hNoDataValues = nullptr;
};
u32 decodeDevice(Context& ctx, const DecodeRequest& rq);

// The same two calls in two halves, for callers that keep several operations in flight on the stream (the asynchronous
// device API, capi.cpp): the enqueue half puts the streaming kernels and the copy of their verdict into `slot` (pinned,
// Context::kAsyncSlotBytes) on the stream and returns true -- or false when the request is not one the streaming kernels
// take blind (then nothing was enqueued); the verdict half is called once the stream has passed that point.
bool encodeEnqueueStreaming(Context& ctx, const EncodeRequest& rq, u8* slot);
// redo: the general path has to repeat the request; else status / sizes are final
void encodeStreamingVerdict(Context& ctx, const EncodeRequest& rq, const u8* slot, bool& redo, u32& status, u32& numBytesNeeded, u32& numBytesWritten);
bool decodeEnqueueStreaming(Context& ctx, const DecodeRequest& rq, u8* slot, u32& epoch);
bool decodeStreamingVerdict(Context& ctx, const u8* slot, u32 epoch, u32* bits = nullptr, int form = -1, int dt = -1);    // true: decoded, checksum good (bits: why not; form: the one that was enqueued, default the last)
// host-pointer calls: streaming kernels + the results' way back to the host enqueued together, one wait (rq holds a device copy of the blob)
u32 decodeSpeculativeToHost(Context& ctx, const DecodeRequest& rq, void* hOut, size_t outBytes, u8* hMask, size_t maskBytes, bool& handled, bool& tried);    // tried: the streaming kernels were enqueued (and may have written pixels)
u32 encodeTilesDevice(Context& ctx, const TilesEncodeRequest& rq, u64& arenaUsed);
u32 decodeTilesDevice(Context& ctx, const TilesDecodeRequest& rq);

// header-only queries (host)
struct BlobInfo
{
  int version = 0, nDepth = 0, nCols = 0, nRows = 0, numValid = 0, nBands = 0, nMasks = 0, nUsesNoData = 0, dt = 0;
  u32 blobSize = 0;
  double zMin = 0, zMax = 0, maxZErr = 0;
};
u32 getBlobInfo(const u8* blob, u32 n, BlobInfo& info, double* mins = nullptr, double* maxs = nullptr, size_t nElem = 0);

// legacy Lerc1 ("CntZImage") blobs: decode only, on the device (lerc1_host.cpp)
bool isLerc1(const u8* hBlob, u32 n);
u32 decodeLerc1(Context& ctx, const DecodeRequest& rq);
u32 lerc1BlobInfo(Context& ctx, const u8* hBlob, u32 n, BlobInfo& info, double* mins, double* maxs, size_t nElem);

}    // namespace lerc
