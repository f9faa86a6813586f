// huffman.h -- 8-bit Huffman image mode (Lerc2::ComputeHuffmanCodes / EncodeHuffman / DecodeHuffman,
// Lerc2.cpp:2270-2606; Huffman.cpp): histograms and bit packing on the device, the 256-entry code
// book on the host.
#pragma once
#include "codec.h"

namespace lerc {

struct HuffmanPlan
{
  bool ok = false;             // a Huffman code book exists (false: fall back to tiling)
  int imageMode = IEM_Tiling;  // IEM_Huffman or IEM_DeltaHuffman
  u32 nBytes = 0;              // table + pixel stream + read-ahead word (Huffman.cpp:85-111)
  std::vector<std::pair<u16, u32> > codes;    // (length, code) per symbol, 256 entries
  std::vector<u8> table;       // serialised code table (Huffman.cpp:126-166)
  u64 nBits = 0;               // bits of the pixel stream
  mutable std::vector<u64> deviceCodes;    // emitHuffman: what the kernels read, kept here until the caller's next synchronisation
};

size_t huffmanScratchBytes(i64 nPix, int nDepth);

// code book + sizes for a plain byte stream with this histogram (the lossless float planes, fpl_EsriHuffman.cpp:238-283);
// false: no code book can be built (fewer than two symbols, or a code longer than 32 bits)
bool planHuffmanFromHisto(const std::vector<int>& histo, HuffmanPlan& plan);


// histograms on the device, code books + sizes on the host; returns false only on a runtime error
// (readyHisto: the 2 x 256 counts, where enqueueHuffmanHisto ran earlier and the stream has been waited for since)
bool planHuffman(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth,
                 int version, HuffmanPlan& plan, const u32* readyHisto = nullptr);
// the histograms only: counts into hHisto[512] once the stream has been waited for (hHisto must live until then)
bool enqueueHuffmanHisto(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth, u32* hHisto);
void enqueueHuffmanHistoDevice(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth, u32* dHisto);    // dHisto: 512 zeroed words, left on the device
// writes table + pixel stream + padding at dOut; enqueues only -- `plan` must outlive the caller's next synchronisation
bool emitHuffman(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth,
                 const HuffmanPlan& plan, u8* dOut, DeviceStatus* dStatus, u8* pin = nullptr, u8* dAlso = nullptr, const u8* pinAlso = nullptr, u32 nAlso = 0);    // pin: pinned host memory, 2048 + plan.table.size() bytes, the caller's until its next synchronisation (or none); dAlso / pinAlso / nAlso: bytes of the caller's that travel with them (the band's header and ranges)
// decodes a Huffman payload starting at blob + dataBegin; returns an ErrCode
u32 decodeHuffman(Context& ctx, int dt, const u8* hBlob, const u8* dBlob, u32 dataBegin, u32 blobEnd, int imageMode,
                  const u8* dMaskBits, int nRows, int nCols, int nDepth, int version, void* dOut, DeviceStatus* dStatus,
                  const u8* bandHead = nullptr, size_t bandHeadLen = 0);    // (the band's first bytes, where the caller holds them already)

}    // namespace lerc
