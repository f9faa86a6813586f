// huffman_dev.h -- types and launch wrappers shared by huffman_host.cpp and huffman_kernels.hip
#pragma once
#include "lerc_common.h"

namespace lerc {

struct HuffGeom { int nRows, nCols, nDepth; };

static const int kHuffRun = 128;         // stream elements per encoder thread (runs with a table of bit offsets)
static const int kHuffSelfRun = 64;      // ... where the packer finds its offsets itself (launchHuffPack with cells)
// 32-bit words per speculative decode sub-sequence (odd; huffSubWords picks).  The decoders are bound by the latency of a
// symbol's two dependent LDS reads, so what counts is how many threads a CU holds -- how little LDS a workgroup takes: its
// slice of the stream (256 sub-sequences) + the table
#ifndef LERC_HUFF_SUB_MIN
#define LERC_HUFF_SUB_MIN 33
#define LERC_HUFF_SUB_MAX 41
#define LERC_HUFF_WG_PER_CU 3
#endif
static const int kHuffSubWordsMin = LERC_HUFF_SUB_MIN, kHuffSubWordsMax = LERC_HUFF_SUB_MAX;
static const int kHuffWgPerCu = LERC_HUFF_WG_PER_CU;    // decoder workgroups a CU holds at that size (LDS)
static const int kHuffLutBits = 12;      // Huffman.h:37 uses the same look-up width

struct HuffDecodeTable
{
  u32 lut[1 << kHuffLutBits];    // (length << 16) | symbol for codes of <= 12 bits, 0xFFFFFFFF otherwise
  int nLong;                     // codes longer than the LUT width, sorted by length
  u32 longCode[256];
  u8 longLen[256];
  u16 longSym[256];
};

void launchHuffHisto(int dt, const void* data, const u8* maskBits, const HuffGeom& g, u32* histos /* 2 x 256, zeroed */, hipStream_t st);
void launchHuffRunBits(int dt, const void* data, const u8* maskBits, const HuffGeom& g, int mode, const u64* codes, u32* runBits,
                       hipStream_t st);
// runBase: bit offset of every run (launchHuffRunBits + launchScan64) -- or, every pixel valid, nullptr and `cells` (one
// zeroed u64 per 256 runs, huffPackCells): the packer finds its offsets itself, in one pass
void launchHuffPack(int dt, const void* data, const u8* maskBits, const HuffGeom& g, int mode, const u64* codes, const u64* runBase,
                    u32* stream /* zeroed; the stream's first byte lies `mis` bytes into stream[0] */, u32 mis, u64* cells, DeviceStatus* status, hipStream_t st);
inline size_t huffPackCells(i64 nElem) { return (size_t)(((nElem + kHuffSelfRun - 1) / kHuffSelfRun + 255) / 256); }
void launchScan64(const u32* in, u64* out /* n + 1 */, u32 n, u64* scratch /* n/256 + 2 */, hipStream_t st);

u32 huffSubWords(u64 streamBits, u32 slots);
void launchHuffInitStarts(u64* starts, u64* prevStarts, u32 nSub, u32 subWords, hipStream_t st);
void launchHuffSync(const u32* stream, u32 mis, u64 nWords, u64 streamBits, const HuffDecodeTable* table, u32 nSub, u32 subWords, u64* starts,
                    u64* prevStarts, u64* exits, u32* counts, u32* bad, bool firstRound, hipStream_t st);
void launchHuffChain(u32 nSub, u64* starts, const u64* exits, u32* changed, hipStream_t st);
void launchValidIndex(const u8* maskBits, const u32* groupBase, i64 nPix, u32* validIdx, hipStream_t st);
void launchHuffEmit(int dt, const u32* stream, u32 mis, u64 nWords, u64 streamBits, const HuffDecodeTable* table, u32 nSub, u32 subWords, const u64* starts,
                    const u64* symBase, const HuffGeom& g, int mode, u64 nSymbols, u32 numValid, const u32* validIdx, bool planar,
                    void* out, hipStream_t st);
// delta mode without a mask and a few values per pixel: symbols decoded plane by plane (planar = [nDepth][nPix] scratch),
// summed and interleaved by launchHuffUndeltaPlanar
bool huffPlanarDecode(int imageMode, const u8* maskBits, int nDepth);
void launchHuffUndeltaPlanar(u8* planar, void* out, const HuffGeom& g, hipStream_t st);
void launchHuffUndelta(int dt, void* data, const u8* maskBits, const HuffGeom& g, hipStream_t st);

}    // namespace lerc
