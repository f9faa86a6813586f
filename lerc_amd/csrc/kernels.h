// kernels.h -- launch wrappers of the HIP kernels (defined in the *.hip files of this directory).
// Every wrapper enqueues on `stream` and returns immediately.
#pragma once
#include "lerc_common.h"

namespace lerc {

// ---- tile_encode.hip ---------------------------------------------------------------------------
// Size-only dry run (Lerc2::WriteTiles with *ppByte == nullptr, Lerc2.cpp:1474-1668): one u32 per
// micro-block position = bytes of all its depth slices.
void launchTileSizes(int dt, int mb, const void* data, const u8* maskBits, const BandParams& p, u32* sizes,
                     DeviceStatus* st, hipStream_t stream);
// Real write: offsets[pos] = byte offset of block position pos relative to `out`.
void launchTileWrite(int dt, int mb, const void* data, const u8* maskBits, const BandParams& p, const u32* offsets,
                     u8* out, DeviceStatus* st, hipStream_t stream);

// ---- tile_decode.hip ---------------------------------------------------------------------------
struct DecodeArgs
{
  const u8* blob;        // start of the band blob (device)
  u32 dataBegin;         // offset of the first block (after header / mask / ranges / flag bytes)
  u32 blobEnd;           // == blobSize of the band
  const u8* maskBits;    // device bit mask or nullptr when all valid
  const double* zMaxVec; // per-depth clamp values (device, nDepth entries)
  const u32* blockOff;   // absolute offset of every sub-block, index = pos * nDepth + iDepth
  const u16* nValidBlk;  // valid pixels per block (device) or nullptr: every block has all its pixels
  void* out;             // decoded pixels
};
void launchTileDecode(int dt, const BandParams& p, const DecodeArgs& a, DeviceStatus* st, hipStream_t stream);

// Block-offset discovery (the stream stores no offsets; SURVEY.md section 7 "hard part 1").
struct WalkPlan
{
  u32 chunkBytes;        // power of two
  u32 window;            // upper bound of a block's byte length for this header
  u32 nChunks;
  u32 nSub;              // number of sub-blocks = nTV * nTH * nDepth
  int uniformN;          // > 0: every block has exactly this many valid pixels; 0: varies (mask / edges)
  u32 candWindow;        // a chunk's candidates: every byte of its first candWindow (one raw block + 1)
  u32 tabled;            // 1: chunks small enough for the kernel that ranks every position (candTab is filled)
  u32 test;              // test knobs (LERC_AMD_TEST_GIVEUP): 8 = D4's first lane distrusts the landings; 16 = the landings are off by a byte
};
struct WalkBuffers
{
  u32* chunkExit;        // [nChunks]   agreed first-block offset of chunk c+1 relative to blob, or ~0u
  u32* chunkEntry;       // [nChunks+1] resolved entry offset of every chunk
  u32* chunkCount;       // [nChunks]   number of sub-blocks that START inside the chunk
  u32* chunkBase;        // [nChunks+1] exclusive scan of chunkCount
  u32* blockOff;         // [nSub]
  const u16* nValidBlk;  // [nTV*nTH]   valid pixels per block position (nullptr when uniformN > 0)
  u32* scratch;          // scan scratch, >= nChunks/1024 + 2 words
  u32* chunkSub;         // [3 * nChunks] where the lowest candidate's chain enters KiB 1, 2, 3 of the chunk (16 bits) | blocks from there to the chunk's exit << 16; ~0u: nowhere
  u32* candTab;          // [nChunks * candWindow] per candidate: where it leaves the chunk (relative, 16 bits) | blocks on the way << 16; 0 = not a block start
};
// ---- legacy Lerc1 z part (lerc1_kernels.hip)
struct Lerc1Geom
{
  int width, height;
  int nTV, nTH;          // numTilesVert / numTilesHori of the part header; a remainder row / column follows when the size does not divide
  u32 tilesAcross, nTiles;
  int allValid;          // the count part is a positive constant (m_bDecoderCanIgnoreMask)
  float maxZInImg;
  double maxZErr;
};
void launchLerc1TileValid(const Lerc1Geom& g, const u8* maskBits, u32* nValid, hipStream_t st);
void launchLerc1Walk(const Lerc1Geom& g, const u8* part, u32 partBytes, const u32* nValid, u32* tileOff /* nTiles + 1 */, DeviceStatus* status,
                     hipStream_t st);
void launchLerc1Decode(int dt, const Lerc1Geom& g, const u8* part, u32 partBytes, const u8* maskBits, const u32* nValid, const u32* tileOff,
                       void* out, DeviceStatus* status, hipStream_t st);

WalkPlan makeWalkPlan(const BandParams& p, u32 dataBegin, u32 blobEnd, int numValid);
void launchWalk(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, DeviceStatus* st,
                hipStream_t stream);
// the same in two parts: the chunk candidates (no mask, no valid counts needed), then the rest
void launchWalkChunks(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, hipStream_t stream);
void launchWalkRest(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, DeviceStatus* st,
                    hipStream_t stream);

// ---- misc_kernels.hip --------------------------------------------------------------------------
// exclusive scan of n u32 values; out[n] receives the total.  scratch >= n/1024 + 2 words.
void launchExclusiveScan(const u32* in, u32* out, u32 n, u32* scratch, hipStream_t stream);

// Fletcher32 partial sums over bytes [begin, begin + len) of `blob`; result words: acc[0] = sum of
// big-endian 16-bit words, acc[1] = position-weighted sum, both mod 65535 (Lerc2.cpp:1037-1064,
// closed form in DESIGN.md).  acc must be zeroed by the caller.
void launchFletcher(const u8* bytes, u32 len, u64* partials /* kFletcherPartials words */, hipStream_t stream);
void launchFletcherPatch(const u64* partials, u32 len, u8* dst /* 4 bytes, little endian */, hipStream_t stream);
void launchFletcherPatchWith(const u64* partials, u32 sums, u32 len, u8* dst, hipStream_t stream);    // partials: the front part; sums: the rest's terms
u32 fletcherFinish(u64 sumWords, u64 sumWeighted, u32 len);

// byte mask (1 = valid) [+ NaN test on float data] -> bit mask, numValid; nValidBlk per block position
void launchBuildMask(int dt, const void* data, const u8* byteMask, int nRows, int nCols, int nDepth, u8* maskBits,
                     BandStats* stats, hipStream_t stream);
// the two in one read of the band, where no TryRaiseMaxZError candidate is wanted (one value a pixel, 16- and 32-bit types, whole words of
// the bit mask): false = not enqueued, the band does not qualify
bool launchMaskStats(int dt, const void* data, const u8* byteMask, int nRows, int nCols, int nDepth, u8* maskBits,
                     u64* mins, u64* maxs, BandStats* stats, hipStream_t stream);
void launchBlockValidCounts(const u8* maskBits, const BandParams& p, u16* nValidBlk, hipStream_t stream);
void launchBitsToBytes(const u8* maskBits, u8* byteMask, i64 nPix, hipStream_t stream);

// per-depth min / max over valid pixels as raw T values (mins[nDepth], maxs[nDepth] device arrays of
// 8-byte slots), plus float diagnostics in BandStats (NaN, all-integer, TryRaiseMaxZError errors)
// raiseMask: bit c set = evaluate TryRaiseMaxZError candidate c (factors 1,2,10,20,100,200,1000,2000,10000).
// mins / maxs hold order-preserving keys (init statKeyInitMin / statKeyInitMax); decode with statKeyTo*.
void launchBandStats(int dt, const void* data, const u8* maskBits, int nRows, int nCols, int nDepth, u32 raiseMask,
                     u64* mins, u64* maxs, BandStats* stats, hipStream_t stream);
u64 statKeyInitMin();
u64 statKeyInitMax();
u64 statKeyToRawBits(int dt, u64 key);
double statKeyToDouble(int dt, u64 key);
// noData values (Lerc.cpp:1241-1552): scan of a private copy of a band (modifies the copy and its byte mask), remap
struct NoDataScan { u64 minKey, maxKey; u32 flags, pad; };    // flags: 1 NaN seen, 2 noData left, 4 mask modified, 8 fractional value
void launchNoDataScan(int dt, void* data, u8* maskBytes, i64 nPix, int nDepth, double orig, NoDataScan* res, hipStream_t stream);
void launchNoDataRemap(int dt, void* data, const u8* maskBytes, const u8* maskBits, i64 nPix, int nDepth, double from, double to, hipStream_t stream);
// bit plane mode (Lerc2.cpp:1071-1229): counts[nDepth * 32 + 1] u32 on the device, see misc_kernels.hip
void launchBitPlaneCounts(int dt, const void* data, const u8* maskBits, int nRows, int nCols, int nDepth, u32* counts, hipStream_t stream);
static const int kFletcherPartials = 2 * 512;    // u64 words written by launchFletcher
void launchMaskGroupCounts(const u8* maskBits, i64 nPix, u32* counts, hipStream_t stream);
// ---- rle_kernels.hip: the run-length coding of a band's validity bits (RLE::compress, RLE.cpp:123-254) on the device.
// out[0 .. *sizeOut) receives the stream with its end marker; *sizeOut = ~0 if it does not fit cap (nothing useful written)
size_t maskRleScratchBytes(size_t nBytes);
void launchMaskRle(const u8* bits, u32 nBytes, u8* out, u32 cap, u32* sizeOut, u8* scratch, hipStream_t stream);
// the other way (rle_kernels.hip): bits[0 .. outBytes) from a stream of rleBytes bytes in device memory
size_t maskRleDecodeScratchBytes(size_t rleBytes);
void launchMaskRleDecode(const u8* rle, u32 rleBytes, u8* bits, u32 outBytes, u8* scratch, DeviceStatus* status, hipStream_t stream);

// raw fallback ("one sweep", Lerc2.cpp:1343-1400): valid pixels copied in order
void launchOneSweep(bool encode, const void* src, void* dst, const u8* maskBits, const u32* groupBase, i64 nPix,
                    int pixelBytes, hipStream_t stream);
void launchFill(void* dst, const void* pixel, int pixelBytes, const u8* maskBits, i64 nPix, hipStream_t stream);
void launchWidenToDouble(int dt, const void* src, double* dst, i64 n, hipStream_t stream);
// small moves by kernels instead of copy commands (misc_kernels.hip): the statistics kernels' inputs set, their results gathered
// in pinned host memory (words of up to five sources back to back), short byte strings out of pinned host memory to their places
void launchStatsInit(u64* mins, u64* maxs, int nDepth, u32* zeroA, u32 nWordsA, u32* zeroB, u32 nWordsB, hipStream_t stream);
void launchWordsGather(const u32* const src[5], const u32 nWords[5], u32* dstPinned, hipStream_t stream);
void launchBytesScatter(u8* const dst[4], const u8* const srcPinned[4], const u32 n[4], hipStream_t stream);

}    // namespace lerc
