// tile_fast.h -- streaming ("fast") kernels for the common case: one band, nDepth == 1, every pixel
// valid, 8 x 8 micro blocks, rows and columns multiples of 8.  Everything else takes the general
// wave-per-block kernels (tile_encode.hip / tile_decode.hip); both produce identical bytes.
#pragma once
#include "lerc_common.h"

namespace lerc {

static const int kFastBlocksPerWG = 64;    // one workgroup = 64 consecutive blocks of the stream (8 rows x 512 cols where a block row is long enough)

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
  u32 stuck;           // != 0: a workgroup gave up waiting for another one (never seen; the host then takes the general path)
  u32 streamSums;      // Fletcher terms of the block stream's bytes at their place in the band: sum of the 16-bit words mod 65535 |
                       // (sum of word index * word mod 65535) << 16 -- a masked band's host adds header and mask to them
};

#ifdef LERC_SMALL_GROUPS                   // (emulator builds: small rasters then take the multi-group hand-offs too)
static const u32 kFastScanGroup = 16, kFastPackGroup = 4;
#else
static const u32 kFastScanGroup = 4096;    // workgroup sizes one workgroup of the scan step takes (1024 threads x 4)
static const u32 kFastPackGroup = 16;      // pack workgroups that add their checksum terms to one accumulator (256 on one address
                                           // made the atomics queue up: k_fast_pack 124 us against 81)
#endif

struct FastBlockDesc      // what pass 1 decided for one block; pass 2 packs from it
{
  u64 mnBits;             // block minimum (raw bits of T)
  u32 w1;                 // nBytes | kind << 16 | tc << 19 | dtRed << 21 | numBits << 24
  u32 pad;
};

// A launch covers nTiles independent rasters of one shape (blockIdx.y = tile; a single raster is nTiles == 1).
// Every per-raster array below then holds nTiles consecutive sets.
struct FastBatch
{
  u32 nTiles;
  u32 nWG;             // workgroups (64 blocks each) per tile
  u64 tileElems;       // pixels from one tile to the next
  u32 nBlobsMore;      // header field: bands that follow this one in the blob (0 for a single band / a tile)
};
LERC_HD u32 fastScanGroups(u32 nWG) { return (nWG + kFastScanGroup - 1u) / kFastScanGroup; }
LERC_HD u32 fastPackGroups(u32 nWG) { return (nWG + kFastPackGroup - 1u) / kFastPackGroup; }
LERC_HD u32 fastTicketStride(u32 nWG) { (void)nWG; return 4u; }
LERC_HD u32 fastWgStride(u32 nWG) { return (nWG + 7u) & ~3u; }    // elements from one tile's wgSize / wgBase set to the next (16-byte aligned)
static const u32 kScanPartWords = 12;
static const int kFastPrefixStage = 128;   // bytes reserved per tile for header + mask count + ranges + mode byte

// One raster (no batch) is encoded in two launches: the statistics step, then a pack step whose first blocks do the scan and
// take the decisions while the others pack (k_fast_pack<SOLO>).  The cells live as long as the codec context.
#ifdef LERC_SMALL_GROUPS                   // (emulator builds: small rasters then take several scan blocks)
static const u32 kSoloSlice = 4;
#else
static const u32 kSoloSlice = 4096;        // workgroups one scan block places (16 per thread)
#endif
struct FastSolo
{
  u64* cells;          // [nWG] epoch (32) | where the workgroup's span starts behind the header (32); nullptr: not this mode
  u32 epoch;
};
// rasters that mode takes: its byte counts are 32 bits wide
LERC_HD bool fastSoloOk(int dt, int nRows, int nCols)
{
  const u64 nBlocks = (u64)((nRows + 7) / 8) * (u64)((nCols + 7) / 8);
  return nBlocks * (1 + 64 * (u64)dtSize(dt)) + 256 < 0xFFFFFFFFull;
}

// One raster, output wanted: ONE launch (k_fast_encode1) -- statistics, block decisions, pack and checksum from a single read
// of the raster.  A workgroup publishes the size of its span in an epoch-tagged cell and adds up the cells of the
// workgroups between the start of the group in front of its own and itself; aggregator blocks (one per kFusedGroup
// workgroups, in front of their group in the grid) leave the bytes in front of each group.
#ifdef LERC_SMALL_GROUPS                   // (emulator builds: small rasters then take several groups)
static const u32 kFusedGroup = 4;
#else
#ifndef LERC_FUSED_GROUP
#define LERC_FUSED_GROUP 256
#endif
static const u32 kFusedGroup = LERC_FUSED_GROUP;    // (a thread reads at most two cells of the window of 2 * kFusedGroup - 1)
#endif
struct FastFused
{
  u64* sizeCell;       // [nWG] epoch (32) | bytes of the workgroup's span (32); nullptr: not this mode
  u64* baseCell;       // [nGroups] epoch (32) | bytes in front of group k (32), k = 1 .. nGroups - 2
  u64* totalCell;      // [nGroups] epoch (32) | bytes of group k (32)
  u64* raise;          // [9] largest first-row rounding error per TryRaiseMaxZError candidate (aggregator 0; read after its arrival)
  u64* packPart;       // [nPackGroups + 1] as k_fast_pack's: A | B << 24 | arrivals << 48 | NaN seen << 53 | non-integer seen << 58; [nPackGroups]: aggregator 0
  u64* keyPart;        // [2 * nPackGroups] largest key, largest complement of a key (zero between calls, like packPart)
  u32 epoch;
  u32 publishEpoch;    // == epoch; a test knob makes it differ, so that nobody ever sees a cell arrive and every waiter gives up
  u32 spinLimit;       // polls before a waiter gives up (2^22; the test knob: a few)
  u32 nWG;             // workgroups per raster (fastFusedUnits units of 64 blocks each)
  // a batch of nTiles rasters of one shape (blockIdx.y = tile): every tile has its own set of the arrays above, cellStride /
  // counterStride words apart, reads its pixels tileElems apart and writes its blob into a slot of its own, outStride bytes apart
  const u8* maskBits;      // != nullptr: the band's validity bits (k_fast_encode1<.., MASKED>)
  u32 payloadAt;           // MASKED: where the block stream begins in `out` (header, mask, ranges and the sweep flag in front of it are the host's)
  u32 nTiles, cellStride, counterStride;
  u64 tileElems, outStride;
  // a batch whose blobs go STRAIGHT into the arena (no slot a tile, no pass that moves them): a tile's last workgroup -- the one that
  // knows the tile's size once its own spans are placed -- claims the tile's room with one atomic add on the batch's cursor and says
  // where in an epoch-tagged cell, the tile's other workgroups wait for that cell before they flush.  Tiles lie in the arena in the
  // order of their claims, 16-byte aligned; `out` is the arena.  nullptr: not this mode
  u64* arenaCursor;        // bytes claimed by the batch so far (zero at launch)
  u64* tileCell;           // [nTiles] epoch (32) | (the tile's offset in the arena) >> 4, 0xFFFFFFFF: the arena is full
  u64* tileOffset;         // [nTiles] out: the tile's offset in the arena, for the host
  u64 arenaBase, arenaCapacity;    // where the batch's first byte goes; the arena's size
};
LERC_HD u32 fastFusedGroups(u32 nWG) { return (nWG + kFusedGroup - 1u) / kFusedGroup; }
// words of a tile's cells (sizes, group bases, group totals, first-row errors) and counters (pack accumulators + aggregator 0, key cells)
LERC_HD u32 fastFusedCellWords(u32 nWG) { return nWG + 2u * ((nWG + kFusedGroup - 1u) / kFusedGroup) + 16u; }
LERC_HD u32 fastFusedCounterWords(u32 nWG) { return 3u * fastPackGroups(nWG) + 1u; }

// consecutive units of 64 blocks a workgroup of k_fast_encode1 takes: as many as usually fit its span image together (two units
// of 32-bit pixels at a ratio of 2.5, three of 16-bit pixels at a ratio of 2), and 32 KB (24 KB) of pixels in flight per workgroup
#ifndef LERC_U32
#define LERC_U32 2
#endif
LERC_HD int fastFusedUnits(int dt) { return dtSize(dt) == 2 ? 3 : LERC_U32; }
LERC_HD u32 fastFusedNumWG(int dt, int nRows, int nCols)
{
  const u64 nUnits = ((u64)((nRows + 7) / 8) * (u64)((nCols + 7) / 8) + 63u) / 64u, per = (u64)fastFusedUnits(dt);
  return (u32)((nUnits + per - 1u) / per);
}

struct FastEncodeBuffers
{
  FastBlockDesc* desc; // [nWG * 64]
  u32* wgSize;         // [nWG + 4] bytes of each workgroup's 64 blocks
  u32* wgBase;         // [nWG + 4] exclusive scan inside a scan group of kFastScanGroup workgroups
  u32* groupBase;      // [nScanGroups + 1] bytes in front of each scan group
  u64* scanPart;       // [kScanPartWords * nScanGroups] what a scan workgroup found: bytes | flags << 32, min key, max key, and (float
                       // types) the largest rounding error of its share of the first raster row per TryRaiseMaxZError candidate
  u64* packPart;       // [nPackGroups (+ 1 one raster: the deciding block)] Fletcher sums of a pack group's workgroups and how many have arrived: A | B << 24 | n << 48
  u32* tickets;        // [fastTicketStride] [0] arrival counter of the scan workgroups
                       // (each kernel zeroes what the next one counts in: the statistics step the scan's ticket, the scan's last
                       // workgroup the pack step's accumulators)
  u64* wgMinKey;       // [nWG] order-preserving key of each workgroup's smallest / largest pixel
  u64* wgMaxKey;       // [nWG]
  u32* wgFlags;        // [nWG] bit 0 NaN seen, bit 1 non-integer value seen
  u8* prefixStage;     // [kFastPrefixStage] the bytes in front of the first block, written by the decide step, copied by the pack step
  u64* tileOffset;     // [nTiles + 1] where each tile's blob starts in the output arena; nullptr: a single raster at offset 0
  FastEncodeResult* result;
  FastSolo solo;
  FastFused fused;
};

// ragged: rows / columns need not be multiples of 8 (the one-launch encoder for a single raster with an output buffer takes such rasters)
bool fastEncodeEligible(int dt, int nRows, int nCols, int nDepth, bool hasMask, double maxZErr, bool ragged = false);
// batches through the one-launch encoder: tile placement (k_fast_tile_offsets) + the move from the tiles' slots into the arena
void launchFastTileCopy(const FastEncodeResult* res, const u64* tileOffset, const u8* slots, u64 slotStride, u64 slotBytes, u8* arena, u32 nTiles,
                        u64 arenaBase, u64 arenaCapacity, hipStream_t st);
u32 fastEncodeNumWG(int nRows, int nCols);
// stage 0: statistics + block sizes (+ first-row rounding errors, float types); 1: scan + decisions + header; 2: pack +
// checksum
// (batches: stage 1 also places the tiles in the arena, from `arenaBase` on; tiles that need the general path take no room)
void launchFastEncode(int stage, const BandParams& assumed, double requestedMaxZErr, u32 raiseCandidates, const void* data, u8* out,
                      u64 outCapacity, u64 arenaBase, const FastEncodeBuffers& b, const FastBatch& batch, hipStream_t st);

// ---- where a workgroup's 64 consecutive blocks lie in the raster -----------------------------------------------
// Block k of the stream is block (k / nTH, k % nTH) of the raster (nTH = nCols / 8 blocks per block row).  Three cases:
// a block row holds a whole number of workgroups (nTH % 64 == 0: one division per workgroup, constant strides inside);
// a workgroup holds a whole number of block rows (nTH a power of two <= 64, e.g. 256 x 256 tiles: shifts and masks);
// anything else (any width that is a multiple of 8): a division per block, the workgroup's blocks wrap around the end of
// a block row wherever it falls, and the last workgroup may hold fewer than 64 blocks (nBlocks).
struct FastSpan { u32 it0, jt0, shift, mask, nTH, nBlocks, k0; };
LERC_HD FastSpan fastSpanOf(u32 wg, u32 nTH, u32 nTV)
{
  FastSpan s;
  const u32 k0 = wg * 64u;
  s.nTH = nTH; s.nBlocks = nTH * nTV; s.k0 = k0;
  const bool pow2 = (nTH & (nTH - 1u)) == 0u;
  if ((nTH & 63u) == 0u) { s.it0 = k0 / nTH; s.jt0 = k0 - s.it0 * nTH; s.shift = 31; s.mask = 0xFFFFFFFFu; }
  else if (pow2 && nTH <= 64u && (s.nBlocks & 63u) == 0u)
  {
    u32 sh = 0;
    while ((1u << (sh + 1)) <= nTH) sh++;
    s.shift = sh; s.it0 = k0 >> sh; s.jt0 = 0; s.mask = nTH - 1u;
  }
  else { s.it0 = k0 / nTH; s.jt0 = k0 - s.it0 * nTH; s.shift = 32; s.mask = 0xFFFFFFFFu; }
  return s;
}
// block j of the workgroup (behind the raster's last block: that last block, so that loads stay inside the raster)
LERC_HD u32 fastSpanRow(const FastSpan& s, u32 j)
{
  if (s.shift == 32u) { const u32 k = (s.k0 + j < s.nBlocks) ? s.k0 + j : s.nBlocks - 1u; return k / s.nTH; }
  return s.it0 + ((s.jt0 + j) >> s.shift);
}
LERC_HD u32 fastSpanCol(const FastSpan& s, u32 j)
{
  if (s.shift == 32u) { const u32 k = (s.k0 + j < s.nBlocks) ? s.k0 + j : s.nBlocks - 1u; return k % s.nTH; }
  return (s.jt0 + j) & s.mask;
}
LERC_HD bool fastSpanHas(const FastSpan& s, u32 j) { return s.k0 + j < s.nBlocks; }    // block j of the workgroup exists
// dimensions the streaming kernels accept: whole 8 x 8 blocks
LERC_HD bool fastDimsOk(int dt, int nRows, int nCols)
{
  (void)dt;
  return nRows > 0 && nCols > 0 && nRows % 8 == 0 && nCols % 8 == 0 && (u64)(nRows / 8) * (u64)(nCols / 8) < 0x7FFFFFC0ull;
}
LERC_HD u32 fastNumWG(int nRows, int nCols) { return (u32)(((u64)((nRows + 7) / 8) * (u64)((nCols + 7) / 8) + 63u) / 64u); }
// any dimensions (the blocks of the last block row / column are smaller): the one-launch encoder
LERC_HD bool fastDimsOkRagged(int nRows, int nCols)
{
  return nRows > 0 && nCols > 0 && (u64)((nRows + 7) / 8) * (u64)((nCols + 7) / 8) < 0x7FFFFFC0ull;
}

// ---- decode side ---------------------------------------------------------------------------------
// The block stream stores no offsets.  Discovery works on 2 KiB chunks of the blob (chunk c = blob bytes
// [c * 2048, (c + 1) * 2048), so every chunk start is 16-byte aligned like the blob itself):
//   k_fast_discover  a workgroup stages kDiscChunks consecutive chunks in LDS (summing their Fletcher32 terms on the way),
//                    finds the bit-stuffed block headers in each chunk's first `window` bytes by their byte pattern, and
//                    lets up to kDiscWalks of them per chunk walk -- in lockstep, listing the block starts they pass --
//                    until they land on a header found in the next chunk's window
//   k_fast_decode    its first blocks resolve: entry of chunk c = the exit all surviving walks of chunk c - 1 agree on; the
//                    walk that starts there is the true path: its block count, scanned, is the index of the chunk's first
//                    block -- left in an epoch-tagged cell per chunk.  The other workgroups decode the blocks that start in
//                    kDecodeChunks chunks each, from the true walks' lists, and check that they tile the stream exactly
#ifndef LERC_CHUNK_BYTES
#define LERC_CHUNK_BYTES 2048
#endif
static const u32 kFastChunkBytes = LERC_CHUNK_BYTES;
static const int kDiscWalks = 8;           // walks per chunk (path heads among the filter's survivors; more: general path)
#ifndef LERC_DISC_CHUNKS
#define LERC_DISC_CHUNKS 16
#endif
static const int kDiscChunks = LERC_DISC_CHUNKS;    // chunks per workgroup of k_fast_discover: 16 (16 threads each; 8 is 8 % slower on one large raster; batches of small blobs: makeFastWalkPlan)
static const int kDiscThreads = (int)((u32)kDiscChunks * kFastChunkBytes / 128u);    // 16 threads per 2 KiB
#ifndef LERC_LIST_CAP
#define LERC_LIST_CAP 128
#endif
static const int kFastListCap = LERC_LIST_CAP;       // block starts a walk can list per chunk (more, i.e. blocks of < 16 bytes on average: general path)
static const u32 kResolveWG = 256;         // threads of a resolving block of k_fast_decode (= of any block of that launch)
#ifdef LERC_SMALL_GROUPS                   // (emulator builds: small streams then take several resolving blocks)
static const u32 kResolveChunks = 4;
#else
static const u32 kResolveChunks = 256;     // chunks a resolving block takes, one per thread
#endif
static const int kOneDiscChunks = (int)(16384u / kFastChunkBytes);       // (kept for the two-launch kernels' templates; no launch uses it any more)
// The one-launch decoder (tile_fast_decode_one.hip): a workgroup stages kOneStage bytes of the blob as sub-chunks of fastOneSub()
// bytes (short walks: half a discovery chunk), walks them all, and decodes the blocks that start in all of them but the
// first -- that one is the last sub-chunk of the workgroup in front, walked again here so that the entry of this workgroup's
// first own sub-chunk (= the exit that sub-chunk's walks agree on) is known without asking anybody.  Workgroup 0 owns its first
// sub-chunk too.  What travels between workgroups is one number: how many blocks start in a workgroup's own bytes (an
// epoch-tagged cell each, and one per kOneGroup workgroups for the group's total).
#ifndef LERC_ONE_STAGE
#define LERC_ONE_STAGE 32768
#endif
static const u32 kOneStage = LERC_ONE_STAGE;               // 32 KiB: three workgroups to a CU (53 KB of LDS each).  24 KiB -- four to a CU, 38 KB
                                                           // each -- is slower, 150 against 117 us on C2: a third more workgroups, and what a
                                                           // workgroup does before its pixels (header, scan, walks, hand-off) costs the same
static const u32 kOneThreads = kOneStage / 64u;            // 16 threads per KiB staged, like k_fast_discover
#ifndef LERC_ONE_SUB
#define LERC_ONE_SUB 1024
#endif
constexpr LERC_HD u32 fastOneSub(int typeBytes) { return (typeBytes == 8 && LERC_ONE_SUB < 1024) ? 1024u : (u32)LERC_ONE_SUB; }    // (at least a raw block + 1: a window lies inside its sub-chunk)
#ifdef LERC_SMALL_GROUPS
static const u32 kOneGroup = 2;
#else
#ifndef LERC_ONE_GROUP
#define LERC_ONE_GROUP 64
#endif
static const u32 kOneGroup = LERC_ONE_GROUP;
#endif
LERC_HD u32 fastOneNumWG(u32 blobBytes, int typeBytes)
{
  const u32 sub = fastOneSub(typeBytes), nch = kOneStage / sub, nSub = (blobBytes + sub - 1u) / sub;
  return nSub <= nch ? 1u : 1u + (nSub - nch + nch - 2u) / (nch - 1u);
}
LERC_HD u32 fastOneGroups(u32 nWG) { return (nWG + kOneGroup - 1u) / kOneGroup; }
LERC_HD u32 fastOneWgStride(u32 bytesBound, int typeBytes) { return (fastOneNumWG(bytesBound, typeBytes) + 3u) & ~1u; }        // cells per tile
LERC_HD u32 fastOneGroupStride(u32 bytesBound, int typeBytes) { return (fastOneGroups(fastOneNumWG(bytesBound, typeBytes)) + 3u) & ~1u; }    // group cells / accumulators per tile
// The scanning decoder (tile_fast_decode_scan.hip), the first tier: a workgroup takes a PIECE of kScanPiece bytes of the blob, with
// one block's length in front of it (the block that ends at the piece's first block begins there) and one behind it (the piece's
// last block ends there).  No walks: every byte is looked at for the count byte of a bit-stuffed block header, and a block start
// is where one candidate begins and another one ends.  The hand-offs are the one-launch decoder's (a cell per workgroup and per
// group of kOneGroup workgroups, the groups' checksum accumulators).
#ifndef LERC_SCAN_PIECE
#ifdef LERC_SMALL_GROUPS
#define LERC_SCAN_PIECE 8192               // (emulator builds: small streams then take several pieces)
#else
#define LERC_SCAN_PIECE 32768
#endif
#endif
static const u32 kScanPiece = LERC_SCAN_PIECE;               // 32 KiB: three workgroups to a CU (51 KB of LDS each)
static const u32 kScanThreads = kScanPiece / 64u;            // 64 bytes a thread
LERC_HD u32 fastScanNumWG(u32 blobBytes) { return blobBytes ? (blobBytes + kScanPiece - 1u) / kScanPiece : 1u; }
// cells per tile for either of the two one-launch decoders
LERC_HD u32 fastAnyWgStride(u32 bytesBound, int typeBytes)
{
  const u32 a = fastOneNumWG(bytesBound, typeBytes), b = fastScanNumWG(bytesBound);
  return ((a > b ? a : b) + 3u) & ~1u;
}
LERC_HD u32 fastAnyGroupStride(u32 bytesBound, int typeBytes)
{
  const u32 a = fastOneNumWG(bytesBound, typeBytes), b = fastScanNumWG(bytesBound);
  return (fastOneGroups(a > b ? a : b) + 3u) & ~1u;
}
#ifndef LERC_DECODE_CHUNKS
#define LERC_DECODE_CHUNKS 4
#endif
static const u32 kDecodeChunks = LERC_DECODE_CHUNKS;        // chunks whose blocks a workgroup of k_fast_decode decodes (divides kResolveWG)
// longest block the streaming walk accepts: the raw form (the reference encoder never emits a longer one)
constexpr u32 kFastWindow(int typeBytes) { return 2u + 64u * (u32)typeBytes; }

// bytes staged in front of a piece of the scanning decoder and behind it (multiples of 32 / 16: bitmap words, 16-byte units)
// (in front of it: TWO blocks' lengths -- the block that ends where the piece's first block begins, and the one in front of that:
// a block of the stream is one that begins where another one ends, and the piece's first block should be told by such a one)
constexpr LERC_HD u32 scanPre(int typeBytes) { return (2u * kFastWindow(typeBytes) + 31u) & ~31u; }
// (behind it: the block that begins with the piece's last byte, and -- a masked band -- one more block's length: blocks that are not
// bit-stuffed behind the piece's end, and the bit-stuffed one behind them, go to the piece in front)
constexpr LERC_HD u32 scanPost(int typeBytes) { return (2u * kFastWindow(typeBytes) + 64u + 15u) & ~15u; }

// sizes the host can bound without reading the blob (grids and buffers); the true values are in FastDecodeParams
struct FastWalkPlan { u32 nChunks, nBlocks, nWaves, discChunks; };    // discChunks: chunks per discovery workgroup of this launch

// what the header parse leaves for the other kernels, and what the host reads back at the end
struct FastDecodeParams
{
  u32 ok;                // the band qualifies for the streaming kernels
  u32 version;
  u32 dataBegin, blobEnd;
  u32 nChunks, nBlocks;
  u32 nTH, nCols, nRows;
  u32 expectChecksum;    // from the header
  u32 checksumOk;        // set by the first resolving block of k_fast_decode
  u32 pad;
  double invScale, zMaxHdr;
};

// what the walks of one chunk found
struct FastChunkRec
{
  u32 exit;                    // the block header of the next chunk's window that all live walks end on (or the blob's end), or ~0
  u32 nLive;
  u16 count[kDiscWalks];       // blocks from walk l's start to `exit`; 0xFFFF: no such walk / it ran into something that is no block
};

// Flags the kernels raise are epoch tagged: cell k == epoch means "raised during this call", so nothing has to be
// cleared between calls (a stale or never written cell matching the epoch by accident only costs a detour through
// the general path).  k: 0 discovery, 1 resolve, 2 gather, 3 decode.
struct FastDecodeBuffers
{
  FastChunkRec* recs;  // [nChunks]
  u16* lists;          // [nChunks * kDiscWalks * kFastListCap] block starts relative to the chunk, per walk
  u64* chunkCell;      // [2 * nChunks] what the resolving blocks found: epoch (32) | index of the chunk's first block (32), and
                       // epoch (32) | the walk that is the true path, 0xFFFF: none (16) | blocks that start in the chunk (16)
  u64* groupCell;      // [ceil(nChunks / kResolveChunks)] epoch (32) | blocks of a resolving block's chunks (32)
  u64* waveFletcher;   // [2 * nWaves] Fletcher partial sums (mod 65535) of the bytes each discovery workgroup staged
  u64* discCell;       // (unused)
  // the one-launch decoder's hand-offs (tile_fast_decode_one.hip)
  u32 wgStride, wgGroupStride;    // cells / group cells (and accumulators) from one tile's set to the next
  u64* wgCell;         // [fastOneWgStride] epoch (32) | where the path of the workgroup's last sub-chunk ends, relative to the next workgroup's first staged byte (16) | blocks of the workgroup's own sub-chunks (16)
  u64* wgGroupCell;    // [fastOneGroupStride] epoch (32) | blocks of group g's workgroups (32), left by the group's last workgroup
  u64* wgAcc;          // [fastOneGroupStride] checksum terms of a group's workgroups and how many have arrived: A | B << 24 | n << 48
                       // (zero between calls: the launch's last workgroup folds and clears them)
  FastDecodeParams* params;   // [nTiles]
  u32* fallback;       // [4 * nTiles] epoch tagged, see above
  FastDecodeParams* hostParams;    // one band: the same two in pinned host memory (written through by the kernels, so that the
  u32* hostFallback;               // verdict needs no copy kernel behind the decode); nullptr for batches
  u32 epoch;
  u32 publishEpoch;    // == epoch; a test knob makes it differ, so that nobody ever sees a cell arrive and every waiter gives up
  u32 spinLimit;       // polls before a waiter gives up (2^22; the test knob: a few)
  u32 scanSpecEnd;     // the scanning decoder: blob bytes the host expects the band to have -- pieces in front of that are asked for without waiting for the header
  u32 scanEarly;       // the scanning decoder: a piece says how many blocks it holds before it has checked them (tile_fast_decode_scan.hip: EARLY)
  u32 scanGridBytes;   // the scanning decoder's launch holds pieces for a blob of so many bytes (0: of the size given).  A decode queued behind
                       // the encode that writes its blob is given a CAPACITY -- as large as the raster -- and a launch sized for that has more
                       // workgroups that find nothing to do than workgroups with a piece; the host sizes the launch like its early loads, by
                       // what the context's last band of this shape had, and a band that turns out larger says so (flag 2) and goes on
  u32 testRewalk;      // test knob (LERC_AMD_TEST_GIVEUP bit 2): the one-launch decoder walks every chunk's path again, as it does
                       // for the rare chunk whose path is not its walk 0
};
// LERC_AMD_TEST_GIVEUP (bit 0: the one-launch encoder, bit 1: the streaming decoder): hand-offs inside a launch never arrive
// -- the path a workgroup takes when it gives up waiting is then the one every call takes (tests/test_gpu_parity.py)
u32 fastTestGiveUp();

// A launch covers nTiles independent blobs of rasters of one shape (blockIdx.y = tile; one raster is nTiles == 1).
// Every buffer above then holds nTiles consecutive slices, sized by the bounds below.
struct FastDecodeBatch
{
  u32 nTiles;
  u32 nChunks, nBlocks, nWaves;      // per tile; nChunks / nWaves are upper bounds (largest blob of the batch)
  u32 discChunks;                    // chunks per discovery workgroup: kDiscChunks, or half of it (see makeFastWalkPlan)
  u64 tileElems;                     // pixels from one tile's output to the next
  const u64* tileOffset;             // device [nTiles]: start of each blob in the arena; nullptr: `blob` itself
  const u32* tileSize;               // device [nTiles]
};
LERC_HD u32 fastChunkStride(u32 nChunks) { return nChunks + 4u; }    // chunk cells per tile
LERC_HD u32 fastGroupStride(u32 nChunks) { return (nChunks + kResolveChunks - 1u) / kResolveChunks + 1u; }    // group cells per tile

bool fastDecodeEligible(int dt, int version, int mb, int nRows, int nCols, int nDepth, bool allValid);
FastWalkPlan makeFastWalkPlan(int nRows, int nCols, u32 sizeGiven, u32 nTiles = 1);
static const int kFastDecodeStages = 2;    // one kernel each: discover (+ header + checksum terms), resolve + decode; stage 2: all of it in one launch
void launchFastDecodeOne(int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                         const FastDecodeBuffers& b, void* out, hipStream_t st);
void launchFastDecode(int stage, int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                      const FastDecodeBuffers& b, void* out, hipStream_t st);
// a masked band's block offsets (tile_fast_decode_scan.hip, MODE 1)
void launchFastScanOffsets(int dt, int nRows, int nCols, const u8* band, u32 version, u32 dataBegin, u32 blobEnd, u32* blockOff, u32 nPos,
                           const FastDecodeBuffers& b, hipStream_t st);
// the scanning decoder: rasters of whole 8 x 8 blocks
bool fastDecodeScanEligible(int nRows, int nCols);
void launchFastDecodeScan(int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                          const FastDecodeBuffers& b, void* out, hipStream_t st);

}    // namespace lerc
