// huffman_host.cpp -- host half of the 8-bit Huffman image mode: the 256-entry code book (built from
// device histograms), its serialisation, and the kernel pipelines for packing / unpacking the pixel
// stream (huffman_kernels.hip).
//
// Reference: Lerc2::ComputeHuffmanCodes (Lerc2.cpp:2270-2307), Huffman::ComputeCodes (Huffman.cpp:35-81),
// ConvertCodesToCanonical (:541-572), GetRange (:383-438), ComputeNumBytesCodeTable (:357-379),
// ComputeCompressedSize (:85-111), WriteCodeTable (:126-166), ReadCodeTable / BitUnStuffCodes (:170-234,
// :471-537), BuildTreeFromCodes (:238-330).
#include "huffman.h"
#include "huffman_dev.h"

#include <algorithm>
#include <cstdio>
#include <queue>

namespace lerc {

namespace {

typedef std::pair<u16, u32> HCode;    // (length, code)

struct HeapItem
{
  int weight;    // minus the symbol count: the std::priority_queue then pops the rarest first
  int node;
  bool operator<(const HeapItem& o) const { return weight < o.weight; }
};

// The reference's code lengths depend on how std::priority_queue orders equal weights; running the
// same container through the same push / pop sequence reproduces them bit for bit.
bool buildCodes(const std::vector<int>& histo, std::vector<HCode>& table)
{
  const int size = (int)histo.size();
  struct Node { int c0, c1, sym; };
  std::vector<Node> nodes;
  std::priority_queue<HeapItem> pq;
  for (int i = 0; i < size; i++)
    if (histo[i] > 0) { nodes.push_back({ -1, -1, i }); pq.push({ -histo[i], (int)nodes.size() - 1 }); }
  if (pq.size() < 2) return false;
  while (pq.size() > 1)
  {
    const HeapItem a = pq.top(); pq.pop();
    const HeapItem b = pq.top(); pq.pop();
    nodes.push_back({ a.node, b.node, -1 });
    pq.push({ a.weight + b.weight, (int)nodes.size() - 1 });
  }
  table.assign(size, HCode(0, 0));
  std::vector<std::pair<int, int> > todo(1, std::make_pair(pq.top().node, 0));
  while (!todo.empty())
  {
    const std::pair<int, int> t = todo.back(); todo.pop_back();
    const Node& nd = nodes[t.first];
    if (nd.c0 < 0) { table[nd.sym].first = (u16)t.second; continue; }
    if (t.second == 32) return false;    // longer than 32 bits: Huffman is abandoned (Huffman.h:84-99)
    todo.push_back(std::make_pair(nd.c0, t.second + 1));
    todo.push_back(std::make_pair(nd.c1, t.second + 1));
  }
  // canonical code assignment: longest codes first, ties by ascending symbol
  std::vector<std::pair<int, u32> > order(size, std::make_pair(0, 0u));
  for (int i = 0; i < size; i++)
    if (table[i].first > 0) order[i] = std::make_pair((int)table[i].first * size - i, (u32)i);
  std::sort(order.begin(), order.end(), [](const std::pair<int, u32>& a, const std::pair<int, u32>& b) { return a.first > b.first; });
  int len = table[order[0].second].first;
  u32 code = 0;
  for (int i = 0; i < size && order[i].first > 0; i++)
  {
    const u32 k = order[i].second;
    const int drop = len - (int)table[k].first;
    code >>= drop;
    len -= drop;
    table[k].second = code++;
  }
  return true;
}

int wrapIdx(int i, int size) { return i - (i < size ? 0 : size); }

// smallest (possibly wrapping) index range that covers all used symbols
bool codeRange(const std::vector<HCode>& t, int& i0, int& i1, int& maxLen)
{
  const int size = (int)t.size();
  if (size == 0 || size >= (1 << 15)) return false;
  int i = 0;
  while (i < size && t[i].first == 0) i++;
  i0 = i;
  i = size - 1;
  while (i >= 0 && t[i].first == 0) i--;
  i1 = i + 1;
  if (i1 <= i0) return false;
  int gapAt = 0, gapLen = 0;
  for (int j = 0; j < size;)
  {
    while (j < size && t[j].first > 0) j++;
    const int k0 = j;
    while (j < size && t[j].first == 0) j++;
    if (j - k0 > gapLen) { gapAt = k0; gapLen = j - k0; }
  }
  if (size - gapLen < i1 - i0) { i0 = gapAt + gapLen; i1 = gapAt + size; }
  if (i1 <= i0) return false;
  int m = 0;
  for (int k = i0; k < i1; k++) m = std::max(m, (int)t[wrapIdx(k, size)].first);
  if (m <= 0 || m > 32) return false;
  maxLen = m;
  return true;
}

// MSB-first bit sink over little-endian 32-bit words
struct WordSink
{
  std::vector<u8>& out;
  u32 cur = 0;
  int used = 0;
  explicit WordSink(std::vector<u8>& o) : out(o) {}
  void flushWord() { for (int i = 0; i < 4; i++) out.push_back((u8)(cur >> (8 * i))); cur = 0; used = 0; }
  void push(u32 v, int len)
  {
    if (32 - used >= len) { cur |= (len == 32) ? v : (v << (32 - used - len)); used += len; if (used == 32) flushWord(); }
    else
    {
      const int rest = len - (32 - used);
      cur |= v >> rest;
      flushWord();
      cur = v << (32 - rest);
      used = rest;
    }
  }
  void finish() { if (used > 0) flushWord(); }
};

void serialiseTable(const std::vector<HCode>& t, std::vector<u8>& out, int lercVersion = kCodecVersion)
{
  out.clear();
  int i0 = 0, i1 = 0, maxLen = 0;
  codeRange(t, i0, i1, maxLen);
  const int size = (int)t.size();
  const int hdr[4] = { 4, size, i0, i1 };
  out.resize(16);
  memcpy(out.data(), hdr, 16);
  // code lengths as a BitStuffer2 "simple" stream (BitStuffer2.cpp:35-75)
  const u32 n = (u32)(i1 - i0);
  const int nb = bitLen((u32)maxLen);
  const int cb = countFieldBytes(n);
  out.push_back((u8)(nb | (((cb == 4) ? 0 : 3 - cb) << 6)));
  for (int i = 0; i < cb; i++) out.push_back((u8)(n >> (8 * i)));
  const size_t at = out.size();
  out.resize(at + ((n * nb + 7) >> 3), 0);
  auto orAt = [&](u32 bitPos, u32 value)    // value's bits into the little-endian bit stream that starts at out[at]
  {
    const u64 v = (u64)value << (bitPos & 7);
    const size_t b = at + (bitPos >> 3);
    for (int k = 0; k < 5 && (v >> (8 * k)); k++) out[b + k] |= (u8)(v >> (8 * k));
  };
  for (u32 i = 0; i < n; i++)
  {
    const u32 len = t[wrapIdx(i0 + (int)i, size)].first;
    if (lercVersion >= 3) orAt(i * nb, len);
    else    // codec 2 bit layout (lerc_common.h: oldBitLayout)
    {
      const OldBitLayout o = oldBitLayout(i, nb, n);
      orAt(o.pos0, len >> o.n1);
      if (o.n1) orAt(o.pos1, len & ((1u << o.n1) - 1u));
    }
  }
  WordSink sink(out);
  for (int i = i0; i < i1; i++)
  {
    const HCode& c = t[wrapIdx(i, size)];
    if (c.first > 0) sink.push(c.second, c.first);
  }
  sink.finish();
}

bool compressedBytes(const std::vector<HCode>& t, const std::vector<int>& histo, u32 tableBytes, u32& nBytes, u64& nBits)
{
  i64 bits = 0, elems = 0;
  for (size_t i = 0; i < histo.size(); i++)
    if (histo[i] > 0) { bits += (i64)histo[i] * t[i].first; elems += histo[i]; }
  if (elems == 0 || bits > (i64)INT_MAX) return false;    // the reference sums the bits in an int
  nBits = (u64)bits;
  nBytes = tableBytes + 4u * (u32)(((((bits + 7) >> 3) + 3) >> 2) + 1);
  return true;
}

// parses the code table at p; `used` = bytes consumed
bool parseTable(const u8* p, size_t n, int lercVersion, std::vector<HCode>& t, size_t& used)
{
  if (lercVersion < 2 || n < 16) return false;
  int hdr[4];
  memcpy(hdr, p, 16);
  if (hdr[0] < 2) return false;
  const int size = hdr[1], i0 = hdr[2], i1 = hdr[3];
  if (i0 >= i1 || i0 < 0 || size < 0 || size > (1 << 15)) return false;
  if (wrapIdx(i0, size) >= size || wrapIdx(i1 - 1, size) >= size) return false;
  size_t at = 16;
  if (n < at + 1) return false;
  const u8 b0 = p[at++];
  const int code = b0 >> 6, cb = (code == 0) ? 4 : 3 - code;
  const int nb = b0 & 31;
  if (cb == 0 || (b0 & 32) || n < at + cb) return false;    // the encoder never uses LUT mode for the lengths
  u32 cnt = 0;
  for (int i = 0; i < cb; i++) cnt |= (u32)p[at + i] << (8 * i);
  at += cb;
  if (cnt != (u32)(i1 - i0)) return false;
  t.assign(size, HCode(0, 0));
  if (nb > 0)
  {
    const size_t nBytes = ((size_t)cnt * nb + 7) >> 3;
    if (n < at + nBytes) return false;
    auto bitsAt = [&](size_t bit, u32 nbits) -> u32
    {
      u64 v = 0;
      for (int k = 0; k < 5; k++) if ((bit >> 3) + k < nBytes) v |= (u64)p[at + (bit >> 3) + k] << (8 * k);
      return (u32)(v >> (bit & 7)) & ((1u << nbits) - 1);
    };
    for (u32 i = 0; i < cnt; i++)
    {
      u32 len;
      if (lercVersion >= 3) len = bitsAt((size_t)i * nb, (u32)nb);
      else
      {
        const OldBitLayout o = oldBitLayout(i, nb, cnt);
        len = bitsAt(o.pos0, o.n0) << o.n1;
        if (o.n1) len |= bitsAt(o.pos1, o.n1);
      }
      t[wrapIdx(i0 + (int)i, size)].first = (u16)len;
    }
    at += nBytes;
  }
  // the codes themselves, MSB first in little-endian words (Huffman.cpp:471-537)
  size_t word = 0;
  int bitPos = 0;
  auto getWord = [&](size_t w, u32& out) -> bool { if (n < at + 4 * (w + 1)) return false; memcpy(&out, p + at + 4 * w, 4); return true; };
  for (int i = i0; i < i1; i++)
  {
    HCode& c = t[wrapIdx(i, size)];
    const int len = c.first;
    if (len == 0) continue;
    if (len > 32) return false;
    u32 w0;
    if (!getWord(word, w0)) return false;
    c.second = (w0 << bitPos) >> (32 - len);
    if (32 - bitPos >= len) { bitPos += len; if (bitPos == 32) { bitPos = 0; word++; } }
    else
    {
      bitPos += len - 32;
      word++;
      u32 w1;
      if (!getWord(word, w1)) return false;
      c.second |= w1 >> (32 - bitPos);
    }
  }
  used = at + 4 * (word + (bitPos > 0 ? 1 : 0));
  return used <= n;
}

bool buildDecodeTable(const std::vector<HCode>& t, HuffDecodeTable& d)
{
  int i0, i1, maxLen;
  if (!codeRange(t, i0, i1, maxLen)) return false;
  for (u32& e : d.lut) e = 0xFFFFFFFFu;
  d.nLong = 0;
  std::vector<std::pair<int, int> > longs;
  for (size_t k = 0; k < t.size(); k++)
  {
    const int len = t[k].first;
    if (len == 0) continue;
    if (len <= kHuffLutBits)
    {
      const u32 base = t[k].second << (kHuffLutBits - len);
      if (base + (1u << (kHuffLutBits - len)) > (1u << kHuffLutBits)) return false;
      for (u32 j = 0; j < (1u << (kHuffLutBits - len)); j++) d.lut[base + j] = ((u32)len << 16) | (u32)k;
    }
    else longs.push_back(std::make_pair(len, (int)k));
  }
  if (longs.size() > 256) return false;
  std::sort(longs.begin(), longs.end());
  for (const std::pair<int, int>& e : longs)
  {
    d.longLen[d.nLong] = (u8)e.first;
    d.longSym[d.nLong] = (u16)e.second;
    d.longCode[d.nLong] = t[e.second].second;
    d.nLong++;
  }
  return true;
}

}    // namespace

size_t huffmanScratchBytes(i64 nPix, int nDepth)
{
  const size_t nElem = (size_t)nPix * nDepth;
  return nElem * 3 + (size_t)nPix * 6 + (2u << 20);
}

bool planHuffmanFromHisto(const std::vector<int>& histo, HuffmanPlan& plan)
{
  plan = HuffmanPlan();
  std::vector<HCode> t;
  std::vector<u8> ser;
  u32 n = 0;
  u64 bits = 0;
  if (!buildCodes(histo, t)) return false;
  serialiseTable(t, ser);
  if (!compressedBytes(t, histo, (u32)ser.size(), n, bits)) return false;
  plan.ok = true;
  plan.imageMode = IEM_Huffman;
  plan.codes = t;
  plan.table = ser;
  plan.nBytes = n;
  plan.nBits = bits;
  return true;
}

bool enqueueHuffmanHisto(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth, u32* hHisto)
{
  hipStream_t st = ctx.activeStream();
  u32* dHisto = ctx.allocT<u32>(512);
  if (!dHisto) return false;
  hipMemsetAsync(dHisto, 0, 512 * 4, st);
  const HuffGeom g{ nRows, nCols, nDepth };
  { ProfScope ps(ctx, "huff_histo"); launchHuffHisto(dt, dData, dMaskBits, g, dHisto, st); }
  return hipMemcpyAsync(hHisto, dHisto, 512 * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
}

// the same without the way home: the counts stay in dHisto (512 words, zeroed by the caller), the caller gathers them
void enqueueHuffmanHistoDevice(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth, u32* dHisto)
{
  const HuffGeom g{ nRows, nCols, nDepth };
  ProfScope ps(ctx, "huff_histo");
  launchHuffHisto(dt, dData, dMaskBits, g, dHisto, ctx.activeStream());
}

bool planHuffman(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth, int version,
                 HuffmanPlan& plan, const u32* readyHisto)
{
  plan = HuffmanPlan();
  u32 hOwn[512];
  const u32* h = readyHisto;
  if (!h)
  {
    if (!enqueueHuffmanHisto(ctx, dt, dData, dMaskBits, nRows, nCols, nDepth, hOwn)) return false;
    if (hipStreamSynchronize(ctx.activeStream()) != hipSuccess) return false;
    h = hOwn;
  }
  std::vector<int> h0(h, h + 256), h1(h + 256, h + 512);

  std::vector<HCode> t0, t1;
  std::vector<u8> s0, s1;
  u32 n0 = 0, n1 = 0;
  u64 bits0 = 0, bits1 = 0;
  if (version >= 4 && buildCodes(h0, t0))
  {
    serialiseTable(t0, s0, version);
    if (!compressedBytes(t0, h0, (u32)s0.size(), n0, bits0)) n0 = 0;
  }
  if (buildCodes(h1, t1))
  {
    serialiseTable(t1, s1, version);
    if (!compressedBytes(t1, h1, (u32)s1.size(), n1, bits1)) n1 = 0;
  }
  // Lerc2.cpp:2289-2306
  bool plain;
  if (n0 > 0 && n1 > 0) plain = (n0 <= n1);
  else if (n0 == 0 && n1 == 0) return true;    // neither works: tiling
  else plain = (n0 > n1);
  plan.ok = true;
  plan.imageMode = plain ? IEM_Huffman : IEM_DeltaHuffman;
  plan.codes = plain ? t0 : t1;
  plan.table = plain ? s0 : s1;
  plan.nBytes = plain ? n0 : n1;
  plan.nBits = plain ? bits0 : bits1;
  return true;
}

bool emitHuffman(Context& ctx, int dt, const void* dData, const u8* dMaskBits, int nRows, int nCols, int nDepth,
                 const HuffmanPlan& plan, u8* dOut, DeviceStatus* dStatus, u8* pin, u8* dAlso, const u8* pinAlso, u32 nAlso)
{
  hipStream_t st = ctx.activeStream();
  const HuffGeom g{ nRows, nCols, nDepth };
  const i64 nElem = (i64)nRows * nCols * nDepth;
  const u32 nRuns = (u32)((nElem + kHuffRun - 1) / kHuffRun);
  const u64 nWords = ((plan.nBits + 31) >> 5) + 1;    // one extra word: the decoder's LUT may read ahead (Lerc2.cpp:2464)

  // the code words and the code table travel through pinned memory (`pin`, the caller's, 2 KB + the table's size: a pageable source
  // is staged by the runtime while the stream waits); the stream is written where the blob has it, behind the table -- `mis` bytes
  // into an aligned word, the packer shifts -- so nothing is copied afterwards
  // (no pinned area handed in -- the lossless float mode's byte planes, several streams a band: the plan's own vectors, which
  // live until the caller's next synchronisation)
  if (!pin) plan.deviceCodes.resize(256);
  u64* hCodes = pin ? reinterpret_cast<u64*>(pin) : plan.deviceCodes.data();
  for (int i = 0; i < 256; i++) hCodes[i] = ((u64)plan.codes[i].first << 32) | plan.codes[i].second;
  const u8* hTable = plan.table.data();
  if (pin) { memcpy(pin + 2048, plan.table.data(), plan.table.size()); hTable = pin + 2048; }
  u64* dCodes = ctx.allocT<u64>(256);
  u32* dRunBits = ctx.allocT<u32>((size_t)nRuns + 4);
  u64* dRunBase = ctx.allocT<u64>((size_t)nRuns + 4);
  u64* dScr = ctx.allocT<u64>((size_t)nRuns / 256 + 8);
  if (!dCodes || !dRunBits || !dRunBase || !dScr) return false;
  u8* dStreamBytes = dOut + plan.table.size();
  const u32 mis = (u32)((uintptr_t)dStreamBytes & 3u);
  u32* dStream = reinterpret_cast<u32*>(dStreamBytes - mis);
  // (zeros first -- from the aligned word the stream begins in, which holds the table's last bytes too -- then the table)
  hipMemsetAsync(dStream, 0, (size_t)mis + (size_t)nWords * 4, st);
  if (pin)    // (one kernel that reads the pinned bytes instead of two copy commands)
  {
    u8* const dst[4] = { reinterpret_cast<u8*>(dCodes), dOut, dAlso, nullptr };
    const u8* const src[4] = { reinterpret_cast<const u8*>(hCodes), hTable, pinAlso, nullptr };
    const u32 n[4] = { 256u * (u32)sizeof(u64), (u32)plan.table.size(), dAlso ? nAlso : 0u, 0u };
    launchBytesScatter(dst, src, n, st);
  }
  else
  {
    hipMemcpyAsync(dCodes, hCodes, 256 * sizeof(u64), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(dOut, hTable, plan.table.size(), hipMemcpyHostToDevice, st);
  }
  if (!dMaskBits && dStatus)
  {
    // every pixel valid: one pass (the packer's workgroups chain their bit counts themselves)
    const size_t nCells = huffPackCells(nElem);
    u64* dCells = ctx.allocT<u64>(nCells + 4);
    if (!dCells) return false;
    hipMemsetAsync(dCells, 0, nCells * 8, st);
    ProfScope ps(ctx, "huff_pack");
    launchHuffPack(dt, dData, nullptr, g, plan.imageMode, dCodes, nullptr, dStream, mis, dCells, dStatus, st);
  }
  else
  {
    { ProfScope ps(ctx, "huff_runbits"); launchHuffRunBits(dt, dData, dMaskBits, g, plan.imageMode, dCodes, dRunBits, st); }
    { ProfScope ps(ctx, "huff_scan"); launchScan64(dRunBits, dRunBase, nRuns, dScr, st); }
    { ProfScope ps(ctx, "huff_pack"); launchHuffPack(dt, dData, dMaskBits, g, plan.imageMode, dCodes, dRunBase, dStream, mis, nullptr, nullptr, st); }
  }
  return hipGetLastError() == hipSuccess;
}

u32 decodeHuffman(Context& ctx, int dt, const u8* hBlob, const u8* dBlob, u32 dataBegin, u32 blobEnd, int imageMode,
                  const u8* dMaskBits, int nRows, int nCols, int nDepth, int version, void* dOut, DeviceStatus* dStatus,
                  const u8* bandHead, size_t bandHeadLen)
{
  (void)dStatus;
  hipStream_t st = ctx.activeStream();
  const HuffGeom g{ nRows, nCols, nDepth };
  const i64 nPix = (i64)nRows * nCols;

  // ---- code table: a few hundred bytes, parsed on the host (out of the bytes the caller fetched with the header where
  // they reach that far: a read from the device costs a wait)
  std::vector<HCode> table;
  size_t used = 0;
  bool parsed = false;
  if (!hBlob && bandHead && bandHeadLen > dataBegin)
    parsed = parseTable(bandHead + dataBegin, std::min<size_t>(bandHeadLen, blobEnd) - dataBegin, version, table, used);
  if (!parsed)
  {
    std::vector<u8> head(std::min<size_t>(blobEnd - dataBegin, 4096));
    if (hBlob) memcpy(head.data(), hBlob + dataBegin, head.size());
    else
    {
      hipMemcpyAsync(head.data(), dBlob + dataBegin, head.size(), hipMemcpyDeviceToHost, st);
      if (hipStreamSynchronize(st) != hipSuccess) return kFailed;
    }
    table.clear(); used = 0;
    if (!parseTable(head.data(), head.size(), version, table, used)) return kFailed;
  }
  // (in pinned memory, behind the call's 64 result bytes: a pageable source is staged by the runtime while the stream waits)
  u8* pinAll = (u8*)ctx.pinned(64 + sizeof(HuffDecodeTable) + 64);
  if (!pinAll) return kFailed;
  HuffDecodeTable* hTab = reinterpret_cast<HuffDecodeTable*>(pinAll + 64);
  if (!buildDecodeTable(table, *hTab)) return kFailed;

  const u32 streamBegin = dataBegin + (u32)used;
  if (streamBegin + 4 > blobEnd) return kFailed;
  const u64 streamBytes = blobEnd - streamBegin;
  const u64 nWords = streamBytes / 4;
  const u64 streamBits = nWords * 32;
  // sub-sequence length: the decoders' workgroups hold ~52 KB of LDS, three of them share a compute unit
  static const int computeUnits = []() { hipDeviceProp_t pr; int dev = 0; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  const u32 subWords = huffSubWords(streamBits, (u32)kHuffWgPerCu * (u32)computeUnits);
  const u64 subBits = (u64)subWords * 32u;
  const u32 nSub = (u32)((streamBits + subBits - 1) / subBits);

  // ---- how many symbols, and which pixel each rank maps to
  u32 numValid = (u32)nPix;
  u32* dValidIdx = nullptr;
  if (dMaskBits)
  {
    const i64 nGroups = (nPix + 31) >> 5;
    u32* dCounts = ctx.allocT<u32>((size_t)nGroups + 4);
    u32* dBase = ctx.allocT<u32>((size_t)nGroups + 4);
    u32* dScr = ctx.allocT<u32>((size_t)nGroups / 1024 + 8);
    if (!dCounts || !dBase || !dScr) return kFailed;
    launchMaskGroupCounts(dMaskBits, nPix, dCounts, st);
    launchExclusiveScan(dCounts, dBase, (u32)nGroups, dScr, st);
    u32* pin = (u32*)ctx.pinned(64);
    if (!pin) return kFailed;
    hipMemcpyAsync(pin, dBase + nGroups, 4, hipMemcpyDeviceToHost, st);
    if (!ctx.sync()) return kFailed;
    numValid = pin[0];
    if (numValid == 0) return kFailed;    // (a band without a valid pixel has no stream; the caller checks the count against the header)
    dValidIdx = ctx.allocT<u32>((size_t)numValid + 4);
    if (!dValidIdx) return kFailed;
    launchValidIndex(dMaskBits, dBase, nPix, dValidIdx, st);
  }
  const u64 nSymbols = (u64)numValid * (u64)nDepth;

  HuffDecodeTable* dTab = ctx.allocT<HuffDecodeTable>(1);
  // the stream is read where the blob has it: `mis` bytes into an aligned word (the kernels put its words together)
  const u32 mis = (u32)((uintptr_t)(dBlob + streamBegin) & 3u);
  const u32* dStream = reinterpret_cast<const u32*>(dBlob + streamBegin - mis);
  u64* dStarts = ctx.allocT<u64>((size_t)nSub + 4);
  u64* dPrev = ctx.allocT<u64>((size_t)nSub + 4);
  u64* dExits = ctx.allocT<u64>((size_t)nSub + 4);
  u32* dCounts = ctx.allocT<u32>((size_t)nSub + 4);
  u64* dSymBase = ctx.allocT<u64>((size_t)nSub + 4);
  u64* dScr64 = ctx.allocT<u64>((size_t)nSub / 256 + 8);
  u32* dFlags = ctx.allocT<u32>(4);
  if (!dTab || !dStarts || !dPrev || !dExits || !dCounts || !dSymBase || !dScr64 || !dFlags) return kFailed;
  hipMemcpyAsync(dTab, hTab, sizeof(HuffDecodeTable), hipMemcpyHostToDevice, st);
  hipMemsetAsync(dFlags, 0, 16, st);
  launchHuffInitStarts(dStarts, dPrev, nSub, subWords, st);

  // ---- synchronise the sub-sequence starts (speculative decode until the chain of exits is stable); the symbol counts
  // are summed right behind every round.  The usual case is a single round, so the second pass -- symbols to their
  // pixels, predictor undone -- is enqueued behind the first round without waiting for its verdict and repeated in the
  // rare case that the chain had to be corrected: one wait, no idle stream in between.
  u32* pin = (u32*)ctx.pinned(64);    // [0] chain changed, [1] bad code seen, [2..3] symbols in the stream
  if (!pin) return kFailed;
  u8* dPlanar = nullptr;
  const bool planar = huffPlanarDecode(imageMode, dMaskBits, nDepth);
  if (planar)
  {
    dPlanar = ctx.allocT<u8>((size_t)nSymbols + 16);
    if (!dPlanar) return kFailed;
  }
  auto secondPass = [&]()
  {
    if (dMaskBits) hipMemsetAsync(dOut, 0, (size_t)nPix * nDepth, st);    // invalid pixels stay 0
    if (planar)
    {
      { ProfScope ps(ctx, "huff_emit"); launchHuffEmit(dt, dStream, mis, nWords, streamBits, dTab, nSub, subWords, dStarts, dSymBase, g, imageMode, nSymbols, numValid, nullptr, true, dPlanar, st); }
      { ProfScope ps(ctx, "huff_undelta"); launchHuffUndeltaPlanar(dPlanar, dOut, g, st); }
      return;
    }
    { ProfScope ps(ctx, "huff_emit"); launchHuffEmit(dt, dStream, mis, nWords, streamBits, dTab, nSub, subWords, dStarts, dSymBase, g, imageMode, nSymbols, numValid, dValidIdx, false, dOut, st); }
    if (imageMode == IEM_DeltaHuffman) { ProfScope ps(ctx, "huff_undelta"); launchHuffUndelta(dt, dOut, dMaskBits, g, st); }
  };
  const int kMaxRounds = 4096;
  int round = 0;
  u64 total = 0;
  for (; round < kMaxRounds; round++)
  {
    { ProfScope ps(ctx, "huff_sync");
      hipMemsetAsync(dFlags, 0, 4, st);
      launchHuffSync(dStream, mis, nWords, streamBits, dTab, nSub, subWords, dStarts, dPrev, dExits, dCounts, dFlags + 1, round == 0, st);
      launchHuffChain(nSub, dStarts, dExits, dFlags, st); }
    { ProfScope ps(ctx, "huff_scan"); launchScan64(dCounts, dSymBase, nSub, dScr64, st); }
    hipMemcpyAsync(pin, dFlags, 8, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(pin + 2, dSymBase + nSub, 8, hipMemcpyDeviceToHost, st);
    if (round == 0) secondPass();
    const bool ok = ctx.sync();
    if (!ok) return kFailed;
    memcpy(&total, pin + 2, 8);
    if (!pin[0]) break;
  }
  if (round == kMaxRounds) { ctx.lastError = "Huffman stream did not synchronise"; return kFailed; }
  if (total < nSymbols) { ctx.lastError = "Huffman stream holds fewer symbols than pixels"; return kFailed; }
  if (round > 0) secondPass();    // (the first one worked from starts that were corrected afterwards)
  return kOk;
}

}    // namespace lerc
