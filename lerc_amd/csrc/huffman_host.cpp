// huffman_host.cpp -- placeholder until the device Huffman path lands (next milestone): no code book
// is produced, so 8-bit lossless bands are written in tiling mode (a valid Lerc2 blob any stock
// decoder accepts) and Huffman blobs are refused by the decoder.
#include "huffman.h"

namespace lerc {

size_t huffmanScratchBytes(i64, int) { return 0; }

bool planHuffman(Context&, int, const void*, const u8*, int, int, int, int, HuffmanPlan& plan)
{
  plan = HuffmanPlan();
  return true;
}

bool emitHuffman(Context&, int, const void*, const u8*, int, int, int, const HuffmanPlan&, u8*, DeviceStatus*) { return false; }

u32 decodeHuffman(Context& ctx, int, const u8*, const u8*, u32, u32, int, const u8*, int, int, int, int, void*, DeviceStatus*)
{
  ctx.lastError = "Huffman image mode is not supported by the device decoder yet";
  return kFailed;
}

}    // namespace lerc
