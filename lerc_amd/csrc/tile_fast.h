// tile_fast.h -- streaming ("fast") kernels for the common case: one band, nDepth == 1, every pixel
// valid, 8 x 8 micro blocks, nRows % 8 == 0 and nCols % 64 == 0.  Everything else takes the general
// wave-per-block kernels (tile_encode.hip / tile_decode.hip); both produce identical bytes.
#pragma once
#include "lerc_common.h"

namespace lerc {

static const int kFastBlocksPerWG = 64;    // one workgroup = 64 consecutive blocks of a block row (8 rows x 512 cols)

// written by the device, read by the host after the single sync of an encode call
struct FastEncodeResult
{
  u32 redo;            // != 0: an assumption of the fast path does not hold, the host must take the general path
  u32 redoReason;      // diagnostic bit set (see tile_fast.hip)
  u32 blobSize;
  u32 nBytesTiling;
  double zMin, zMax;
  u64 minKey, maxKey;
  u32 prefixLen;       // bytes before the first block
  u32 checksum;
};

static const int kFastSlots = 64;          // atomics of the workgroups are spread over this many global slots

struct FastBlockDesc      // what pass 1 decided for one block; pass 2 packs from it
{
  u64 mnBits;             // block minimum (raw bits of T)
  u32 w1;                 // nBytes | kind << 16 | tc << 19 | dtRed << 21 | numBits << 24
  u32 pad;
};

struct FastEncodeBuffers
{
  FastBlockDesc* desc; // [nWG * 64]
  u32* wgSize;         // [nWG] bytes of each workgroup's 64 blocks
  u32* wgBase;         // [nWG + 1] exclusive scan
  u64* slotMinKey;     // [kFastSlots]
  u64* slotMaxKey;     // [kFastSlots]
  u32* slotFlags;      // [kFastSlots] bit 0 NaN seen, bit 1 non-integer value seen
  u64* slotFletcher;   // [2 * kFastSlots] Fletcher partial sums of the bytes the workgroups wrote
  u32* scanScratch;
  double* row0RaiseErr;    // [9] TryRaiseMaxZError rounding errors of the first row (float types), or nullptr
  FastEncodeResult* result;
};

bool fastEncodeEligible(int dt, int nRows, int nCols, int nDepth, bool hasMask, double maxZErr);
u32 fastEncodeNumWG(int nRows, int nCols);
// stage 0: statistics + block sizes; 1: scan + decisions + header; 2: pack + Fletcher sums; 3: checksum patch
void launchFastEncode(int stage, const BandParams& assumed, double requestedMaxZErr, u32 raiseCandidates, const void* data, u8* out,
                      u32 outCapacity, const FastEncodeBuffers& b, hipStream_t st);

// ---- decode side ---------------------------------------------------------------------------------
static const u32 kFastChunkBytes = 4096;
static const u32 kFastSubBytes = 512;      // the walk also records the first block start behind every sub-chunk boundary
static const int kFastSubPerChunk = (int)(kFastChunkBytes / kFastSubBytes);
// longest block the streaming walk accepts: the raw form (the reference encoder never emits a longer one)
constexpr u32 kFastWindow(int typeBytes) { return 2u + 64u * (u32)typeBytes; }

struct FastWalkPlan { u32 nChunks, nBlocks; };

struct FastDecodeBuffers
{
  u32* chunkExit;      // [nChunks]
  u16* countAt;        // [nChunks * window] #blocks from a surviving start to the end of its chunk, 0xFFFF elsewhere
  u32* chunkEntry;     // [nChunks]
  u32* chunkCount;     // [nChunks]
  u32* chunkBase;      // [nChunks + 1]
  u32* subEntry;       // [nChunks * kFastSubPerChunk] agreed first block start at / behind a sub-chunk boundary, or ~0
  u64* slotFletcher;   // [2 * kFastSlots]
  u64* fletcherOut;    // [2]
  u32* scanScratch;
  u32* fallback;       // != 0: the general path must redo the band
};

bool fastDecodeEligible(int dt, int version, int mb, int nRows, int nCols, int nDepth, bool allValid);
FastWalkPlan makeFastWalkPlan(int nRows, int nCols, u32 dataBegin, u32 blobEnd);
// stage 0: chunk walk; 1: resolve + scan; 2: decode + Fletcher sums
void launchFastDecode(int stage, const BandParams& p, const FastWalkPlan& wp, const u8* blob, u32 dataBegin, u32 blobEnd,
                      const FastDecodeBuffers& b, void* out, DeviceStatus* status, hipStream_t st);

}    // namespace lerc
