// fpl_host.cpp -- host half of the lossless float / double image mode.
//
// Reference: LosslessFPCompression::ComputeHuffmanCodesFltSlice / EncodeHuffmanFlt / DecodeHuffmanFltSlice
// (fpl_Lerc2Ext.cpp:456-606, :391-421, :737-866), selectInitialLinearOrCrossDelta (:337-389), generateListOfTestBlocks
// (:57-101), getBestLevel2 (:237-322), fpl_Compression::getEntropySize (fpl_Compression.cpp:85-113),
// fpl_EsriHuffman::EncodeHuffman / DecodeHuffman (fpl_EsriHuffman.cpp:306-560).
//
// The reference's choices (predictor, difference order per plane, coding of each plane) hang on sums of entropy
// estimates computed in double precision from byte histograms; the histograms come off the device as integers and the
// sums are formed here in the reference's order with the same libm calls, so the choices are the same.
#include "fpl.h"
#include "fpl_dev.h"

#include <cmath>
#include <cstring>
#include <string>

namespace lerc {

namespace {

// fpl_Compression.cpp:85-113
long entropyBytes(const u32* histo)
{
  int total = 0;
  for (int i = 0; i < 256; i++) total += (int)histo[i];
  double totalBits = 0;
  for (int i = 0; i < 256; i++)
  {
    if (histo[i] == 0) continue;
    const unsigned long cnt = histo[i];
    const double p = (double)total / cnt;
    const double bits = log2(p);
    totalBits += (bits * cnt);
  }
  return (long)((totalBits + 7) / 8);
}

// fpl_Lerc2Ext.cpp:57-101 -- rows [top, top + height) of the raster, as element spans
void testBlocks(int width, int height, std::vector<FplSpan>& out)
{
  const size_t size = (size_t)width * height;
  const int target = 8 * 1024;
  const double t = round((double)size / target);
  int count = (int)round(sqrt(t + 1));
  int blockHeight = target / width;
  if (blockHeight < 4) blockHeight = 4;
  while ((count * blockHeight > height) && (count > 1)) count--;
  const float topMargin = (float)((height - count * blockHeight) / (2.0 * count));
  const float delta = 2.0f * topMargin + blockHeight;
  for (int i = 0; i < count; i++)
  {
    long top = (long)(topMargin + delta * i);
    long h = blockHeight;
    if (top < 0) top = 0;
    if (top + h > height) h = height - top;
    if (h > 0) out.push_back(FplSpan{ (i64)top * width, (i64)h * width });
  }
}

// fpl_Lerc2Ext.cpp:239-267
void snippets(size_t size, std::vector<FplSpan>& out)
{
  const unsigned int target = 1024 * 8;
  const double t = round((double)size / target);
  int count = (int)round(sqrt(t + 1));
  while (count * target > size && (count > 0)) count--;
  if (count <= 0) return;    // (the reference divides by zero here and ends up with no snippets as well)
  const float topMargin = (float)(((int)size - count * target) / (2.0 * count));
  const float delta = 2.0f * topMargin + target;
  for (int i = 0; i < count; i++)
  {
    long start = (long)(topMargin + delta * i);
    int len = (int)target;
    if (start < 0) start = 0;
    if (start + len > (int)size) len = (int)size - (int)start;
    if (len > 0) out.push_back(FplSpan{ (i64)start, (i64)len });
  }
}

FplGeom geomOf(int unit, int nRows, int nCols, int nDepth)
{
  FplGeom g;
  g.unit = unit;
  g.nElem = (i64)nRows * nCols * nDepth;
  g.cols = (nDepth == 1) ? nCols : nDepth;                  // fpl_Lerc2Ext.cpp:432-452
  g.rows = (nDepth == 1) ? nRows : (i64)nCols * nRows;
  g.nanToZero = (nDepth == 1) ? 1 : 0;
  return g;
}

bool waitFor(Context& ctx) { return ctx.sync(); }

// log2((double)total / c), c = 1 .. total, by this host's libm -- the numbers fpl_Compression.cpp:98-104 works with.
// The sample counts of test blocks and snippets repeat from call to call, so the few tables are kept.
const std::vector<double>& log2Table(u32 total)
{
  static thread_local std::vector<std::pair<u32, std::vector<double> > > kept;
  for (auto& e : kept) if (e.first == total) return e.second;
  if (kept.size() >= 8) kept.erase(kept.begin());
  kept.emplace_back(total, std::vector<double>((size_t)total + 1, 0.0));
  std::vector<double>& t = kept.back().second;
  for (u32 c = 1; c <= total; c++) t[c] = log2((double)total / (unsigned long)c);
  return t;
}

static const u32 kTableMax = 1u << 16;    // entries over all tables; more (test blocks of > 450 000 elements): the host does the sums

// The entropy estimates of nHist histograms on the device -> pinned host array (valid after the next sync), or nullptr
// if the sample counts do not fit the tables (then the caller reads the histograms back and uses entropyBytes)
const i64* enqueueEntropies(Context& ctx, const u32* dHistos, size_t nHist, const std::vector<FplSpan>& spans)
{
  hipStream_t st = ctx.activeStream();
  FplEntropyTables tb;
  memset(&tb, 0, sizeof(tb));
  std::vector<double> image;
  for (const FplSpan& sp : spans)
  {
    const u32 total = (u32)((sp.len + kFplPrime - 1) / kFplPrime);
    bool have = false;
    for (u32 t = 0; t < tb.nTables; t++) have = have || tb.total[t] == total;
    if (have || total == 0) continue;
    if (tb.nTables == 4 || image.size() + total + 1 > kTableMax) return nullptr;
    const std::vector<double>& t = log2Table(total);
    tb.total[tb.nTables] = total; tb.at[tb.nTables] = (u32)image.size(); tb.nTables++;
    image.insert(image.end(), t.begin(), t.end());
  }
  double* dTables = ctx.allocT<double>(image.size() + 1);
  i64* dOut = ctx.allocT<i64>(nHist);
  i64* hOut = (i64*)ctx.pinned(nHist * sizeof(i64));
  if (!dTables || !dOut || !hOut) return nullptr;
  if (!image.empty()) hipMemcpyAsync(dTables, image.data(), image.size() * sizeof(double), hipMemcpyHostToDevice, st);    // (pageable: staged before the call returns)
  launchFplEntropy(dHistos, (u32)nHist, dTables, tb, dOut, st);
  hipMemcpyAsync(hOut, dOut, nHist * sizeof(i64), hipMemcpyDeviceToHost, st);
  return hOut;
}

}    // namespace

size_t fplEncodeScratchBytes(i64 nElem, int unit)
{
  const size_t n = (size_t)nElem;
  const size_t nSpans = (size_t)sqrt((double)n / 8192 + 2) + 4;
  return n * unit + (size_t)unit * (size_t)fplPlaneStride(nElem)        // units, planes
    + 3 * (n / 4096 + 64) * 4 + (1u << 20)                                  // PackBits carries, entropy tables
    + n + n / 8 + (4u << 20)                                                 // Huffman stream + run tables of one plane
    + nSpans * (6 * 8 * 256 * 4 + 8 * kFplLevels * 256 * 4 + 64) + (1u << 20);
}

size_t fplDecodeScratchBytes(i64 nElem, int unit)
{
  const size_t n = (size_t)nElem;
  return (size_t)unit * (size_t)fplPlaneStride(nElem) + 6 * n + n * unit / 64 + (8u << 20);
}

bool planLosslessFloat(Context& ctx, int dt, const void* dData, const u8* dByteMask, bool nansFiltered, int nRows, int nCols, int nDepth,
                       FplPlan& plan)
{
  hipStream_t st = ctx.activeStream();
  plan = FplPlan();
  const int U = (dt == DT_Double) ? 8 : 4;
  FplGeom g = geomOf(U, nRows, nCols, nDepth);
  if (nansFiltered) g.nanToZero = 0;
  plan.unit = U; plan.nElem = g.nElem; plan.nRows = nRows; plan.nCols = nCols; plan.nDepth = nDepth;
  if (g.nElem <= 0 || g.nElem > (i64)INT_MAX) return false;
  const u32 n = (u32)g.nElem;

  // ---- 1. predictor: entropy estimates over the test blocks, for no / row / row + column differences (:337-389)
  std::vector<FplSpan> blocks;
  testBlocks((int)g.cols, (int)g.rows, blocks);
  if (blocks.empty()) return false;
  const size_t h1Len = (size_t)3 * 2 * U * 256;
  FplSpan* dBlocks = ctx.allocT<FplSpan>(blocks.size());
  u32* dH1 = ctx.allocT<u32>(blocks.size() * h1Len);
  if (!dBlocks || !dH1) return false;
  hipMemcpyAsync(dBlocks, blocks.data(), blocks.size() * sizeof(FplSpan), hipMemcpyHostToDevice, st);
  size_t est[3] = { 0, 0, 0 };
  {
    ProfScope ps(ctx, "fpl_predictor_samples");
    launchFplPredictorSamples(dData, dByteMask, g, dBlocks, (u32)blocks.size(), dH1, st);
    // histogram index: (block * 6 + predictor * 2 + plain / differenced) * U + byte
    const i64* ent = enqueueEntropies(ctx, dH1, blocks.size() * 6 * U, blocks);
    if (ent)
    {
      if (!waitFor(ctx)) return false;
      for (int p = 0; p < 3; p++)
        for (size_t blk = 0; blk < blocks.size(); blk++)
          for (int b = 0; b < U; b++)
          {
            const i64 plain = ent[(blk * 6 + (size_t)p * 2 + 0) * U + b], prime = ent[(blk * 6 + (size_t)p * 2 + 1) * U + b];
            if (plain < 0 || prime < 0) { ctx.lastError = "lossless float: entropy table missing"; return false; }
            est[p] += (size_t)std::min(plain, prime);
          }
    }
    else
    {
      // (histograms come back through the context's pinned mirror: a pageable target costs a staging copy at ~1 GB/s)
      const u32* h1 = (const u32*)ctx.pinned(blocks.size() * h1Len * 4);
      if (!h1) return false;
      hipMemcpyAsync((void*)h1, dH1, blocks.size() * h1Len * 4, hipMemcpyDeviceToHost, st);
      if (!waitFor(ctx)) return false;
      for (int p = 0; p < 3; p++)
        for (size_t blk = 0; blk < blocks.size(); blk++)
          for (int b = 0; b < U; b++)
          {
            const u32* hp = &h1[blk * h1Len + ((size_t)(p * 2 + 0) * U + b) * 256];
            const u32* hq = &h1[blk * h1Len + ((size_t)(p * 2 + 1) * U + b) * 256];
            const size_t plain = (size_t)entropyBytes(hp), prime = (size_t)entropyBytes(hq);
            est[p] += std::min(plain, prime);
          }
    }
  }
  int predictor = 0;
  for (int p = 1; p < 3; p++) if (est[p] < est[predictor]) predictor = p;
  plan.predictor = predictor;

  // ---- 2. the predicted units, then the extra difference order of every byte plane (:237-322)
  void* dUnits = ctx.alloc((size_t)n * U);
  if (!dUnits) return false;
  { ProfScope ps(ctx, "fpl_predict"); launchFplPredict(dData, dByteMask, g, predictor, dUnits, st); }
  const int maxDelta = kFplMaxDelta - predictor;    // Predictor::getMaxByteDelta
  std::vector<FplSpan> snips;
  snippets(n, snips);
  FplLevels lv;
  memset(&lv, 0, sizeof(lv));
  if (!snips.empty())
  {
    const size_t h2Len = (size_t)kFplLevels * 256;
    FplSpan* dSnips = ctx.allocT<FplSpan>(snips.size());
    u32* dH2 = ctx.allocT<u32>((size_t)U * snips.size() * h2Len);
    if (!dSnips || !dH2) return false;
    hipMemcpyAsync(dSnips, snips.data(), snips.size() * sizeof(FplSpan), hipMemcpyHostToDevice, st);
    ProfScope ps(ctx, "fpl_level_samples");
    launchFplLevelSamples(dUnits, g, dSnips, (u32)snips.size(), dH2, st);
    // histogram index: (byte * snippets + snippet) * levels + level
    const i64* ent = enqueueEntropies(ctx, dH2, (size_t)U * snips.size() * kFplLevels, snips);
    const u32* h2 = nullptr;
    if (!ent)
    {
      h2 = (const u32*)ctx.pinned((size_t)U * snips.size() * h2Len * 4);
      if (!h2) return false;
      hipMemcpyAsync((void*)h2, dH2, (size_t)U * snips.size() * h2Len * 4, hipMemcpyDeviceToHost, st);
    }
    if (!waitFor(ctx)) return false;
    for (int b = 0; b < U; b++)
    {
      size_t best = 0;
      int ret = 0;
      for (int l = 0; l <= maxDelta; l++)
      {
        size_t comp = 0;
        for (size_t s = 0; s < snips.size(); s++)
        {
          const size_t hi = ((size_t)b * snips.size() + s) * kFplLevels + l;
          if (ent && ent[hi] < 0) { ctx.lastError = "lossless float: entropy table missing"; return false; }
          comp += ent ? (size_t)ent[hi] : (size_t)entropyBytes(&h2[hi * 256]);
        }
        if (comp < best || l == 0) { best = comp; ret = l; }
        else break;
      }
      lv.level[b] = ret;
    }
  }

  // ---- 3. the planes as they get coded, their histograms
  const i64 stride = fplPlaneStride(g.nElem);
  plan.dPlanes = ctx.allocT<u8>((size_t)U * stride);
  const size_t h3Len = (size_t)U * 256 + 8 + 8;    // histograms, "equals its successor" counts, PackBits sizes
  u32* dH3 = ctx.allocT<u32>(h3Len);
  if (!plan.dPlanes || !dH3) return false;
  hipMemsetAsync(dH3, 0, h3Len * 4, st);
  { ProfScope ps(ctx, "fpl_symbols"); launchFplSymbols(dUnits, g, lv, plan.dPlanes, dH3, st); }
  std::vector<u32> h3(h3Len);
  {
    u32* pin = (u32*)ctx.pinned(h3Len * 4);
    if (!pin) return false;
    hipMemcpyAsync(pin, dH3, h3Len * 4, hipMemcpyDeviceToHost, st);
    if (!waitFor(ctx)) return false;
    memcpy(h3.data(), pin, h3Len * 4);
  }

  // code books; PackBits is sized only where it can win
  bool needPackBits[8] = { false, false, false, false, false, false, false, false }, anyPackBits = false;
  for (int b = 0; b < U; b++)
  {
    FplPlanePlan& pp = plan.plane[b];
    pp.level = lv.level[b];
    std::vector<int> histo(256);
    int distinct = 0, firstSym = 0;
    for (int i = 0; i < 256; i++) { histo[i] = (int)h3[(size_t)b * 256 + i]; if (histo[i] > 0 && distinct++ == 0) firstSym = i; }
    if (distinct < 2) { pp.mode = 1; pp.value = (u8)firstSym; pp.size = 6; continue; }
    if (!planHuffmanFromHisto(histo, pp.huff)) { ctx.lastError = "lossless float: no Huffman code book for a byte plane"; return false; }
    // every run of equal bytes emits at least one byte, every 128 single bytes one count byte more; at most `eq` runs are
    // longer than one byte, so at least n - 2 eq are single
    const u64 eq = h3[(size_t)U * 256 + b], runs = (u64)n - eq, singles = (2 * eq < n) ? (u64)n - 2 * eq : 0;
    needPackBits[b] = runs + singles / 128 < std::min<u64>(pp.huff.nBytes, n);
    anyPackBits = anyPackBits || needPackBits[b];
  }
  if (anyPackBits)
  {
    const size_t mark = ctx.used();
    PackBitsBuffers pb;
    pb.carry1 = ctx.allocT<u32>(packBitsCarryCount(n));
    pb.carry2 = ctx.allocT<u32>(packBitsCarryCount(n));
    pb.carry3 = ctx.allocT<u32>(packBitsCarryCount(n));
    u32* dPbSize = dH3 + (size_t)U * 256 + 8;
    if (!pb.carry1 || !pb.carry2 || !pb.carry3) return false;
    for (int b = 0; b < U; b++)
    {
      if (!needPackBits[b]) continue;
      ProfScope ps(ctx, "fpl_packbits_size");
      launchPackBitsPlan(plan.dPlanes + (size_t)b * stride, n, pb, st);
      hipMemcpyAsync(dPbSize + b, packBitsSize(pb, n), 4, hipMemcpyDeviceToDevice, st);
    }
    u32* pin = (u32*)ctx.pinned(64);
    if (!pin) return false;
    hipMemcpyAsync(pin, dPbSize, 8 * 4, hipMemcpyDeviceToHost, st);
    if (!waitFor(ctx)) return false;
    memcpy(h3.data() + (size_t)U * 256 + 8, pin, 8 * 4);
    ctx.rewind(mark);
  }

  // ---- 4. how every plane is coded (fpl_EsriHuffman.cpp:306-381)
  u32 total = 1;    // predictor code
  for (int b = 0; b < U; b++)
  {
    FplPlanePlan& pp = plan.plane[b];
    if (pp.mode != 1)
    {
      const long numBytes = (long)pp.huff.nBytes, rle = needPackBits[b] ? (long)h3[(size_t)U * 256 + 8 + b] : 0;
      if (rle > 0 && rle < numBytes && rle < (long)n) { pp.mode = 3; pp.size = (u32)rle + 1; }
      else if (numBytes >= (long)n) { pp.mode = 2; pp.size = n + 1; }
      else { pp.mode = 0; pp.size = (u32)numBytes + 1; }
    }
    if ((u64)total + pp.size + 6 > (u64)INT_MAX) return false;
    total += pp.size + 6;
  }
  plan.nBytes = total;
  return true;
}

bool emitLosslessFloat(Context& ctx, const FplPlan& plan, u8* dOut)
{
  hipStream_t st = ctx.activeStream();
  const int U = plan.unit;
  const u32 n = (u32)plan.nElem;
  const i64 stride = fplPlaneStride(plan.nElem);
  // small pieces (predictor code, plane headers, mode bytes) from one host image
  std::vector<u8> small(1 + (size_t)U * 16);
  std::vector<std::pair<size_t, std::pair<size_t, size_t> > > copies;    // (offset in blob, (offset in small, length))
  size_t at = 0, sm = 0;
  small[sm] = (u8)plan.predictor; copies.push_back(std::make_pair(at, std::make_pair(sm, (size_t)1))); at += 1; sm += 1;
  struct Body { size_t at; int b; };
  std::vector<Body> bodies;
  for (int b = 0; b < U; b++)
  {
    const FplPlanePlan& pp = plan.plane[b];
    const size_t s0 = sm;
    small[sm++] = (u8)b;            // byte_index
    small[sm++] = (u8)pp.level;     // best_level
    memcpy(&small[sm], &pp.size, 4); sm += 4;
    small[sm++] = (u8)pp.mode;
    size_t len = 7;
    if (pp.mode == 1) { small[sm++] = pp.value; memcpy(&small[sm], &n, 4); sm += 4; len += 5; }
    copies.push_back(std::make_pair(at, std::make_pair(s0, len)));
    bodies.push_back(Body{ at + 7, b });
    at += 6 + pp.size;
  }
  u8* pin = (u8*)ctx.pinned(small.size());
  if (!pin) return false;
  memcpy(pin, small.data(), small.size());
  for (const auto& c : copies) hipMemcpyAsync(dOut + c.first, pin + c.second.first, c.second.second, hipMemcpyHostToDevice, st);

  // (the one-pass Huffman packer reports here if it ever gives up on its look-back; allocated in front of the planes'
  // scratch, which is rewound plane by plane)
  DeviceStatus* dPackStatus = ctx.allocT<DeviceStatus>(1);
  if (!dPackStatus) return false;
  hipMemsetAsync(dPackStatus, 0, sizeof(DeviceStatus), st);
  for (const Body& body : bodies)
  {
    const FplPlanePlan& pp = plan.plane[body.b];
    const u8* dPlane = plan.dPlanes + (size_t)body.b * stride;
    u8* dst = dOut + body.at;
    const size_t mark = ctx.used();
    if (pp.mode == 2) hipMemcpyAsync(dst, dPlane, n, hipMemcpyDeviceToDevice, st);
    else if (pp.mode == 3)
    {
      PackBitsBuffers pb;
      pb.carry1 = ctx.allocT<u32>(packBitsCarryCount(n));
      pb.carry2 = ctx.allocT<u32>(packBitsCarryCount(n));
      pb.carry3 = ctx.allocT<u32>(packBitsCarryCount(n));
      if (!pb.carry1 || !pb.carry2 || !pb.carry3) return false;
      ProfScope ps(ctx, "fpl_packbits_emit");
      launchPackBitsPlan(dPlane, n, pb, st);
      launchPackBitsEmit(dPlane, n, pb, dst, st);
    }
    else if (pp.mode == 0)
    {
      // (the planes are plain byte streams: the 8-bit image mode's packer in its "no predictor" form)
      if (!emitHuffman(ctx, DT_Byte, dPlane, nullptr, plan.nRows, plan.nCols, plan.nDepth, pp.huff, dst, dPackStatus)) return false;
    }
    ctx.rewind(mark);
  }
  DeviceStatus hs;
  memset(&hs, 0, sizeof(hs));
  hipMemcpyAsync(&hs, dPackStatus, sizeof(hs), hipMemcpyDeviceToHost, st);
  if (!ctx.sync()) return false;    // the pinned image must outlive the copies
  if (hs.error) { ctx.lastError = "lossless float: the Huffman packer gave up"; return false; }
  return true;
}

namespace {
u32 failAt(Context& ctx, int where)
{
  if (ctx.lastError.empty()) ctx.lastError = "lossless float stream: check " + std::to_string(where) + " failed";
  return kFailed;
}
}    // namespace

u32 decodeLosslessFloat(Context& ctx, int dt, const u8* hBand, const u8* dBand, u32 dataBegin, u32 blobEnd, int nRows, int nCols,
                        int nDepth, void* dOut)
{
  hipStream_t st = ctx.activeStream();
  const int U = (dt == DT_Double) ? 8 : 4;
  const FplGeom g = geomOf(U, nRows, nCols, nDepth);
  if (g.nElem <= 0 || g.nElem > (i64)INT_MAX) return failAt(ctx, 1);
  const u32 n = (u32)g.nElem;
  const i64 stride = fplPlaneStride(g.nElem);
  auto fetch = [&](u32 off, size_t len, u8* dst) -> bool
  {
    if ((u64)off + len > blobEnd) return false;
    if (hBand) { memcpy(dst, hBand + off, len); return true; }
    u8* pin = (u8*)ctx.pinned(64);
    if (!pin || len > 64) return false;
    if (hipMemcpyAsync(pin, dBand + off, len, hipMemcpyDeviceToHost, st) != hipSuccess || !ctx.sync()) return false;
    memcpy(dst, pin, len);
    return true;
  };

  u32 at = dataBegin;
  u8 predCode = 0;
  if (!fetch(at, 1, &predCode) || predCode > 2) return failAt(ctx, 2);    // fpl_Lerc2Ext.cpp:756-760
  at += 1;
  u8* dPlanes = ctx.allocT<u8>((size_t)U * stride);
  u32* dScan = ctx.allocT<u32>((size_t)n / 1024 + 16);
  if (!dPlanes || !dScan) return failAt(ctx, 3);
  int byteIndex[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  struct PendingCheck { u32* dResult; };
  std::vector<u32*> walkResults;
  for (int b = 0; b < U; b++)
  {
    u8 hd[12];
    if (!fetch(at, 6, hd)) return failAt(ctx, 4);
    const u32 bi = hd[0], level = hd[1];
    u32 size = 0;
    memcpy(&size, hd + 2, 4);
    if (bi >= (u32)U || level > (u32)kFplMaxDelta) return failAt(ctx, 5);
    at += 6;
    if ((u64)at + size > blobEnd || size < 1) return failAt(ctx, 6);
    byteIndex[b] = (int)bi;
    u8* dPlane = dPlanes + (size_t)b * stride;
    u8 mode = 0;
    if (!fetch(at, 1, &mode)) return failAt(ctx, 7);
    const size_t mark = ctx.used();
    if (mode == 1)
    {
      u8 rl[5];
      if (size < 6 || !fetch(at + 1, 5, rl)) return failAt(ctx, 8);
      u32 cnt = 0;
      memcpy(&cnt, rl + 1, 4);
      if (cnt != n) return failAt(ctx, 9);
      hipMemsetAsync(dPlane, rl[0], n, st);
    }
    else if (mode == 2)
    {
      if ((u64)size < (u64)n + 1) return failAt(ctx, 10);
      hipMemcpyAsync(dPlane, dBand + at + 1, n, hipMemcpyDeviceToDevice, st);
    }
    else if (mode == 3)
    {
      const u32 nIn = size - 1;
      u8* scratch = (u8*)ctx.alloc(nIn ? packBitsDecodeScratchBytes(nIn) : 256);
      u32* dRes = ctx.allocT<u32>(4);
      if (!scratch || !dRes || nIn == 0) return failAt(ctx, 100);
      ProfScope ps(ctx, "fpl_packbits_decode");
      launchPackBitsDecode(dBand + at + 1, nIn, n, scratch, dRes, dPlane, st);
      u32* pin = (u32*)ctx.pinned(64);
      if (!pin) return failAt(ctx, 12);
      hipMemcpyAsync(pin, dRes, 8, hipMemcpyDeviceToHost, st);
      if (!ctx.sync()) return failAt(ctx, 13);
      if (pin[1] != 1u) { ctx.lastError = "lossless float: damaged PackBits plane"; return failAt(ctx, 14); }
    }
    else if (mode == 0)
    {
      const u32 rc = decodeHuffman(ctx, DT_Byte, hBand, dBand, at + 1, at + size, IEM_Huffman, nullptr, nRows, nCols, nDepth, 5, dPlane, nullptr);
      if (rc != kOk) return rc;
    }
    else return failAt(ctx, 15);
    // restoreSequence (:128-165): the byte-wise differences, highest order first, each a running sum from its own start
    for (int l = (int)level; l > 0; l--)
      if ((u32)l < n)
      {
        ProfScope ps(ctx, "fpl_byte_sums");
        launchBytePrefixSum(dPlane + (l - 1), n - (u32)(l - 1), dScan, st);
      }
    if (!ctx.sync()) return failAt(ctx, 16);    // scratch of this plane is handed back below
    ctx.rewind(mark);
    at += size;
  }

  const bool cross = (predCode == 2);
  { ProfScope ps(ctx, "fpl_gather"); launchFplGather(dPlanes, byteIndex, g, predCode == 0, dOut, st); }
  if (cross)
  {
    const u32 nSeg = fplColumnSegments(g.rows);
    void* dPartial = ctx.alloc((size_t)nSeg * (size_t)g.cols * U);
    if (!dPartial) return failAt(ctx, 17);
    ProfScope ps(ctx, "fpl_column_sums");
    launchFplColumnSums(dOut, g, dPartial, nSeg, st);
  }
  if (predCode != 0) { ProfScope ps(ctx, "fpl_row_sums"); launchFplRowSums(dOut, g, st); }
  return kOk;
}

}    // namespace lerc
