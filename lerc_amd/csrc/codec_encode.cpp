// codec_encode.cpp -- lerc_encode() pipeline on device-resident pixels.
//
// Host logic mirrors Lerc::EncodeInternal (Lerc.cpp:628-789: band loop, mask reuse, flags) and
// Lerc2::ComputeNumBytesNeededToWrite / Lerc2::Encode (Lerc2.cpp:179-480: mode decision, section
// order).  Every sweep over pixels is a HIP kernel; the host only sees a few scalars per band.
#include "codec.h"
#include "huffman.h"
#include "fpl.h"
#include "tile_fast.h"
#include <cfloat>
#include <functional>
#include <future>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>

namespace lerc {

namespace {

struct MaskState    // what the reference keeps inside its Lerc2 object between bands
{
  u32 fplStale = 0;    // bytes of lossless float planes that were coded but not written (see encodeBand)
  bool allValid = true;
  int numValid = 0;
  u8* dBits = nullptr;          // device bit mask (valid when !allValid)
  std::vector<u8> hBits;        // host copy, for RLE and band-to-band comparison
};

struct Sync
{
  hipStream_t s;
  bool wait() const { return hipStreamSynchronize(s) == hipSuccess; }
};

bool isIntegral(double z) { return z == floor(z + 0.5); }

}    // namespace

// ------------------------------------------------------------------------------------------------
// noData values: what Lerc::FilterNoDataAndNaN (float, Lerc.cpp:1378-1552) and its integer sibling (:1241-1374)
// decide once the sweep over the band (k_nodata_scan) is done.  In: the requested error bound and noData value;
// out: the bound to encode with, whether the blob has to carry a noData value, and the value it is remapped to.
// ------------------------------------------------------------------------------------------------
struct NoDataDecision
{
  bool active = false;       // this band came with a noData value
  bool needNoData = false;   // some valid pixels keep noData in some depths: the header carries the value
  bool allInt = false;       // float types: header flag bIsInt
  bool remap = false;
  double maxZErr = 0, noDataOrig = 0, noDataNew = 0, remapFrom = 0, remapTo = 0;
  const u8* dData = nullptr; // the filtered copy of the band
  const u8* dMask = nullptr; // ... and of its byte mask
  bool modifiedMask = false;
  bool empty = false;        // no value left at all
};

template<class T> static bool isIntValT(T z) { return z == (T)floor((double)z + 0.5); }    // Lerc.h:271

// Lerc.cpp:1558-1618
template<class T>
static bool findNoDataBelowMin(double minVal, double maxZErr, bool allInt, double lowIntLimit, T& out)
{
  std::vector<T> cand;
  if (allInt)
  {
    const double dist[] = { 4 * maxZErr, 1, 10, 100, 1000, 10000 };
    for (double d : dist) cand.push_back((T)(minVal - d));
    cand.push_back((T)(minVal > 0 ? floor(minVal / 2) : minVal * 2));
    std::sort(cand.begin(), cand.end(), std::greater<double>());
    for (T v : cand)
      if ((v > (T)lowIntLimit) && (v < (T)(minVal - 2 * maxZErr)) && isIntValT(v)) { out = v; return true; }
  }
  else
  {
    const double dist[] = { 4 * maxZErr, 0.0001, 0.001, 0.01, 0.1, 1, 10, 100, 1000, 10000 };
    for (double d : dist) cand.push_back((T)(minVal - d));
    cand.push_back((T)(minVal > 0 ? minVal / 2 : minVal * 2));
    std::sort(cand.begin(), cand.end(), std::greater<double>());
    const T lowest = (T)(std::is_same<T, float>::value ? -FLT_MAX : -DBL_MAX);
    for (T v : cand)
      if ((v > lowest) && (v < (T)(minVal - 2 * maxZErr))) { out = v; return true; }
  }
  return false;
}

template<class T>
static u32 decideNoDataFloat(const NoDataScan& sc, bool any, double minVal, double maxVal, int nDepth, double maxZErr, double noDataValue,
                             NoDataDecision& d)
{
  const bool isF32 = std::is_same<T, float>::value;
  const T origNoData = (T)noDataValue;
  const bool noDataLeft = (sc.flags & 2u) != 0;
  bool allInt = !(sc.flags & 8u);
  const double lowInt = isF32 ? -(double)(1L << 23) : -(double)((i64)1 << 53), highInt = -lowInt;
  d.maxZErr = maxZErr; d.noDataNew = noDataValue;
  if (!any) { d.empty = true; d.maxZErr = 0; return kOk; }
  d.needNoData = noDataLeft;
  (void)nDepth;
  double e = maxZErr;
  if (allInt)
  {
    allInt = allInt && (minVal >= lowInt) && (minVal <= highInt) && (maxVal >= lowInt) && (maxVal <= highInt);
    if (noDataLeft) allInt = allInt && isIntValT(origNoData) && (origNoData >= lowInt) && (origNoData <= highInt);
    if (allInt) e = std::max(0.5, floor(maxZErr));
  }
  d.allInt = allInt;
  if (e == 0) { d.maxZErr = maxZErr; return kOk; }
  {
    const double dist = allInt ? floor(e) : 2 * e;
    if ((origNoData >= minVal - dist) && (origNoData <= maxVal + dist)) { d.maxZErr = allInt ? 0.5 : 0; return kOk; }
  }
  if (noDataLeft)
  {
    T remap = origNoData;
    if (findNoDataBelowMin<T>(minVal, e, allInt, lowInt, remap))
    {
      if (remap != origNoData) { d.remap = true; d.remapFrom = (double)origNoData; d.remapTo = (double)remap; d.noDataNew = (double)remap; }
    }
    else if ((double)origNoData >= minVal) e = allInt ? 0.5 : 0;
  }
  d.maxZErr = e;
  return kOk;
}

template<class T>
static u32 decideNoDataInt(const NoDataScan& sc, bool any, double minVal, double maxVal, double lo, double hi, double maxZErr,
                           double noDataValue, NoDataDecision& d)
{
  const T orig = (T)noDataValue;
  d.needNoData = (sc.flags & 2u) != 0;
  d.noDataNew = noDataValue;
  double e = std::max(0.5, floor(maxZErr));
  const double dist = floor(e);
  if (!any) { d.empty = true; d.maxZErr = 0.5; return kOk; }
  if (((double)orig >= minVal - dist) && ((double)orig <= maxVal + dist)) { d.maxZErr = 0.5; return kOk; }
  if (d.needNoData)
  {
    const double minDist = floor(e) + 1;
    double remap = minVal - minDist;
    T nd = orig;
    if (remap >= lo) nd = (T)remap;
    else
    {
      e = 0.5;
      remap = minVal - 1;
      if (remap >= lo) nd = (T)remap;
      else
      {
        remap = maxVal + 1;
        if ((remap <= hi) && (remap < (double)orig)) nd = (T)remap;
      }
    }
    if (nd != orig) { d.remap = true; d.remapFrom = (double)orig; d.remapTo = (double)nd; d.noDataNew = (double)nd; }
  }
  d.maxZErr = e;
  return kOk;
}

// sweeps a private copy of band iBand and fills `d`; workspace comes from ctx (behind whatever is allocated so far)
static u32 filterNoData(Context& ctx, const EncodeRequest& rq, int iBand, NoDataDecision& d)
{
  hipStream_t st = ctx.activeStream();
  const int dt = rq.dt, nD = rq.nDepth;
  const int tb = dtSize(dt);
  const i64 nPix = (i64)rq.nRows * rq.nCols, nElem = nPix * nD;
  const double noData = rq.hNoDataValues[iBand];
  static const double tlo[6] = { -128, 0, -32768, 0, -2147483648.0, 0 }, thi[6] = { 127, 255, 32767, 65535, 2147483647.0, 4294967295.0 };
  if (dt == DT_Float && (noData < -FLT_MAX || noData > FLT_MAX)) return kWrongParam;
  if (dt < DT_Float && (noData < tlo[dt] || noData > thi[dt])) return kWrongParam;
  if (noData != noData) return kWrongParam;
  u8* dCopy = ctx.allocT<u8>((size_t)nElem * tb + 256);
  u8* dMask = ctx.allocT<u8>((size_t)nPix + 256);
  NoDataScan* dScan = ctx.allocT<NoDataScan>(1);
  if (!dCopy || !dMask || !dScan) return kFailed;
  const u8* src = (const u8*)rq.dData + (size_t)iBand * nElem * tb;
  hipMemcpyAsync(dCopy, src, (size_t)nElem * tb, hipMemcpyDeviceToDevice, st);
  if (rq.nMasks > 0) hipMemcpyAsync(dMask, rq.dValidBytes + ((rq.nMasks > 1) ? (size_t)iBand * nPix : 0), (size_t)nPix, hipMemcpyDeviceToDevice, st);
  else hipMemsetAsync(dMask, 1, (size_t)nPix, st);
  launchNoDataScan(dt, dCopy, dMask, nPix, nD, noData, dScan, st);
  NoDataScan sc;
  hipMemcpyAsync(&sc, dScan, sizeof(sc), hipMemcpyDeviceToHost, st);
  if (hipStreamSynchronize(st) != hipSuccess) return kFailed;
  const bool any = sc.minKey != statKeyInitMin() || sc.maxKey != statKeyInitMax();
  const double minVal = any ? statKeyToDouble(dt, sc.minKey) : 0, maxVal = any ? statKeyToDouble(dt, sc.maxKey) : 0;
  d = NoDataDecision();
  d.active = true; d.noDataOrig = noData; d.dData = dCopy; d.dMask = dMask; d.modifiedMask = (sc.flags & 4u) != 0;
  u32 rc = kOk;
  switch (dt)
  {
    case DT_Float:  rc = decideNoDataFloat<float>(sc, any, minVal, maxVal, nD, rq.maxZErr, noData, d); break;
    case DT_Double: rc = decideNoDataFloat<double>(sc, any, minVal, maxVal, nD, rq.maxZErr, noData, d); break;
    case DT_Char:   rc = decideNoDataInt<signed char>(sc, any, minVal, maxVal, tlo[dt], thi[dt], rq.maxZErr, noData, d); break;
    case DT_Byte:   rc = decideNoDataInt<unsigned char>(sc, any, minVal, maxVal, tlo[dt], thi[dt], rq.maxZErr, noData, d); break;
    case DT_Short:  rc = decideNoDataInt<short>(sc, any, minVal, maxVal, tlo[dt], thi[dt], rq.maxZErr, noData, d); break;
    case DT_UShort: rc = decideNoDataInt<unsigned short>(sc, any, minVal, maxVal, tlo[dt], thi[dt], rq.maxZErr, noData, d); break;
    case DT_Int:    rc = decideNoDataInt<int>(sc, any, minVal, maxVal, tlo[dt], thi[dt], rq.maxZErr, noData, d); break;
    default:        rc = decideNoDataInt<unsigned int>(sc, any, minVal, maxVal, tlo[dt], thi[dt], rq.maxZErr, noData, d); break;
  }
  if (rc != kOk) return rc;
  if (d.remap) launchNoDataRemap(dt, dCopy, dMask, nullptr, nPix, nD, d.remapFrom, d.remapTo, st);
  return kOk;
}

// ------------------------------------------------------------------------------------------------
// streaming kernels: buffers + launches for nTiles rasters of one shape (a single raster is nTiles == 1)
// ------------------------------------------------------------------------------------------------
struct FastEncodeLaunch
{
  FastEncodeBuffers fb;
  FastBatch batch;
  BandParams bp;
  u32 cand;
  double maxZErr;
};

// LERC_AMD_MASKED_STREAMING=0: masked bands keep the general kernels for their block stream (a test / tuning knob)
static bool maskedStreamingOn()
{
  static const bool on = []() { const char* e = getenv("LERC_AMD_MASKED_STREAMING"); return !(e && atoi(e) == 0); }();
  return on;
}

// LERC_AMD_ENCODE_LAUNCHES=2 keeps the two-launch form (statistics, then scan + pack) for a single raster: a tuning / test knob
static bool fastEncodeOneLaunch()
{
  static const bool one = []() { const char* e = getenv("LERC_AMD_ENCODE_LAUNCHES"); return !(e && e[0] == '2'); }();
  return one;
}

static u32 encodeBand(Context& ctx, const EncodeRequest& rq, int iBand, MaskState& ms, std::vector<u8>& prevByteValid,
                      bool& anyMaskModified, u8* dBandOut, u32 capacityLeft, u32& bandBytes)
{
  // (LERC_AMD_HOST_TIMES: where the host is, microseconds into the band -- a tuning aid)
  static const bool kTL = getenv("LERC_AMD_HOST_TIMES") != nullptr;
  const auto tl0 = std::chrono::steady_clock::now();
  auto TL = [&](const char* what) { if (kTL) fprintf(stderr, "  [tl] %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl0).count(), what); };

  hipStream_t st = ctx.activeStream();
  Sync sync{ st };
  const int dt = rq.dt, nD = rq.nDepth, nCols = rq.nCols, nRows = rq.nRows;
  const int tb = dtSize(dt);
  const i64 nPix = (i64)nRows * nCols, nElem = nPix * nD;
  const bool isFlt = dt >= DT_Float;
  const u8* dData = (const u8*)rq.dData + (size_t)iBand * nElem * tb;
  const u8* dByteMask = (rq.nMasks > 0) ? rq.dValidBytes + ((rq.nMasks > 1) ? (size_t)iBand * nPix : 0) : nullptr;
  bandBytes = 0;
  // a noData value: the band is filtered into a private copy first (pixels that are noData throughout leave the
  // mask, the value may move below the data range), and the decisions below come from that filter
  NoDataDecision nd;
  if (rq.hUsesNoData && rq.hUsesNoData[iBand])
  {
    const u32 rc = filterNoData(ctx, rq, iBand, nd);
    if (rc != kOk) return rc;
    dData = nd.dData;
    dByteMask = nd.dMask;
  }

  // ---- device scratch of this band
  DeviceStatus* dStatus = ctx.allocT<DeviceStatus>(1);
  BandStats* dStats = ctx.allocT<BandStats>(1);
  BandStats* dStatsRow0 = ctx.allocT<BandStats>(1);    // (the first row's TryRaiseMaxZError errors, measured beside the mask's statistics)
  u64* dMins = ctx.allocT<u64>(nD);
  u64* dMaxs = ctx.allocT<u64>(nD);
  u8* dNewBits = ctx.allocT<u8>((size_t)((nPix + 7) >> 3) + 16);
  if (!dStatus || !dStats || !dStatsRow0 || !dMins || !dMaxs || !dNewBits) return kFailed;
  hipMemsetAsync(dStatus, 0, sizeof(DeviceStatus), st);
  hipMemsetAsync(dStats, 0, sizeof(BandStats), st);

  struct HostRes { BandStats stats; BandStats row0; DeviceStatus status; };
  std::vector<u64> hMins(nD), hMaxs(nD);
  HostRes hr;

  // ---- 1. validity: caller's byte mask, minus pixels that are NaN in every depth (Lerc.cpp:1440-1476)
  bool bandAllValid = true;
  int bandNumValid = (int)nPix;
  bool modifiedMask = false;
  bool haveBits = false;    // dNewBits holds this band's bit mask
  // single band: the bits travel to pinned host memory as soon as they are final, and a helper thread codes their RLE
  // (a millisecond for the 8 MB of an 8192 x 8192 mask) while this thread goes on launching and waiting for kernels
  const u8* bitsOnTheWay = nullptr;
  size_t nBitsOnTheWay = 0;
  std::future<std::vector<u8> > rleFuture;
  // ... or, a large mask: coded on the device (rle_kernels.hip) -- the bits never leave it, the stream's size and its first
  // kRleFirst bytes travel home beside the statistics kernels, and codeMask() below picks them up
  // (LERC_AMD_DEVICE_RLE=0: never; =<bytes>: from masks of that many bytes on -- a test knob; default: 256 KB, as for the helper threads)
  static const size_t kDeviceRleFrom = []() -> size_t { const char* e = getenv("LERC_AMD_DEVICE_RLE"); const long v = e ? atol(e) : 1; return v <= 0 ? ~(size_t)0 : v < 16 ? (size_t)(256u << 10) : (size_t)v; }();
  static const u32 kRleCap = 4u << 20, kRleFirst = 64u << 10;
  bool deviceRle = false;
  u8* dRle = nullptr;
  u32* pinRle = nullptr;        // [0]: the stream's size (~0: it did not fit kRleCap), from byte 16 on: its first kRleFirst bytes
  size_t nBitsOnDevice = 0;
  auto sendBitsHome = [&]() -> bool
  {
    const size_t nb = (size_t)((nPix + 7) >> 3);
    if (nb >= kDeviceRleFrom && nb < 0xFFFFFFF0ull)
    {
      u8* scratch = ctx.allocT<u8>(maskRleScratchBytes(nb));
      dRle = ctx.allocT<u8>((size_t)kRleCap + 64);
      u32* dSize = ctx.allocT<u32>(4);
      pinRle = (u32*)ctx.pinnedAux((size_t)kRleCap + 64);
      if (scratch && dRle && dSize && pinRle)
      {
        hipStream_t side = ctx.forkSide();
        hipStream_t sr = side ? side : st;
        ProfScope ps(ctx, "mask_rle");
        launchMaskRle(dNewBits, (u32)nb, dRle, kRleCap, dSize, scratch, sr);
        hipMemcpyAsync(pinRle, dSize, 4, hipMemcpyDeviceToHost, sr);
        hipMemcpyAsync((u8*)pinRle + 16, dRle, kRleFirst, hipMemcpyDeviceToHost, sr);
        hipEventRecord(ctx.auxEvent(), sr);
        deviceRle = true; nBitsOnDevice = nb;
        return true;
      }
    }
    u8* pin = (u8*)ctx.pinnedAux(nb);
    if (!pin) return false;
    // (beside the stream, so that the statistics kernels do not wait behind 8 MB on their way over PCIe: a third of a millisecond)
    hipStream_t side = nb >= (256u << 10) ? ctx.forkSide() : nullptr;
    hipEvent_t ev = ctx.auxEvent();
    hipMemcpyAsync(pin, dNewBits, nb, hipMemcpyDeviceToHost, side ? side : st);
    hipEventRecord(ev, side ? side : st);
    bitsOnTheWay = pin; nBitsOnTheWay = nb;
    if (nb < (256u << 10)) return true;    // small masks: codeMask() does it in line (starting a thread costs ~30 us)
    try
    {
      rleFuture = std::async(std::launch::async, [pin, nb, ev]()
      {
        std::vector<u8> out;
        const auto t0 = std::chrono::steady_clock::now();
        // (the workers are started now, while the bits travel, and wait for them one by one)
        if (!rleEncodeWhenReady(pin, nb, [ev]() { return hipEventSynchronize(ev) == hipSuccess; }, out)) out.clear();
        if (getenv("LERC_AMD_HOST_TIMES")) fprintf(stderr, "  [tl] helper: %zu -> %zu bytes, %.1f us after its start\n", nb, out.size(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        return out;    // (empty: the copy failed; an RLE stream is never empty)
      });
    }
    catch (...) { rleFuture = std::future<std::vector<u8> >(); }    // no thread to be had: codeMask() codes the mask in line
    return true;
  };
  // TryRaiseMaxZError's candidates whose error bound beats the request (Lerc2.cpp:1244-1253) are pruned on the raster's first
  // row; where a mask is made first, that row is measured in the same wait (every wait costs the stream ~50 us of idling)
  static const double errCand[9] = { 1, 0.5, 0.1, 0.05, 0.01, 0.005, 0.001, 0.0005, 0.0001 };
  static const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
  u32 candAll = 0;
  if (isFlt && rq.maxZErr > 0)
    for (int c = 0; c < 9; c++) if (errCand[c] / 2 > rq.maxZErr) candAll |= 1u << c;
  bool row0Measured = false;
  bool maskStats = false;    // the mask's kernel has made the band's statistics as well (launchMaskStats)
  auto buildMask = [&]() -> bool
  {
    // (inputs set and results gathered by kernels, through pinned memory: see runStats)
    static_assert(sizeof(BandStats) % 4 == 0, "words");
    const u32 wStats = (u32)(sizeof(BandStats) / 4);
    u32* pin = (u32*)ctx.pinned((size_t)(2u * wStats + 4u + 16u) * 4u);
    if (!pin) return false;
    launchStatsInit(dMins, dMaxs, nD, reinterpret_cast<u32*>(dStats), wStats, candAll ? reinterpret_cast<u32*>(dStatsRow0) : nullptr, candAll ? wStats : 0u, st);
    // (the band's statistics in the same read, where the band qualifies: they stand if no TryRaiseMaxZError candidate survives the first row)
    maskStats = false;
    if (dByteMask && !nd.active) { ProfScope ps(ctx, "mask_stats"); maskStats = launchMaskStats(dt, dData, dByteMask, nRows, nCols, nD, dNewBits, dMins, dMaxs, dStats, st); }
    if (!maskStats) { ProfScope ps(ctx, "build_mask"); launchBuildMask(dt, dData, dByteMask, nRows, nCols, nD, dNewBits, dStats, st); }
    row0Measured = false;
    if (candAll)
    {
      { ProfScope ps(ctx, "band_stats_row0"); launchBandStats(dt, dData, dNewBits, 1, nCols, nD, candAll, dMins, dMaxs, dStatsRow0, st); }    // (with the bits just made: all ones if nothing is invalid)
      row0Measured = true;
    }
    {
      const u32* const src[5] = { reinterpret_cast<const u32*>(dStats), candAll ? reinterpret_cast<const u32*>(dStatsRow0) : nullptr,
                                  maskStats ? reinterpret_cast<const u32*>(dMins) : nullptr, maskStats ? reinterpret_cast<const u32*>(dMaxs) : nullptr, nullptr };
      const u32 nw[5] = { wStats, candAll ? wStats : 0u, maskStats ? 2u : 0u, maskStats ? 2u : 0u, 0u };
      launchWordsGather(src, nw, pin, st);
    }
    if (!sync.wait()) return false;
    memcpy(&hr.stats, pin, sizeof(BandStats));
    if (candAll) memcpy(&hr.row0, pin + wStats, sizeof(BandStats));
    if (maskStats) { memcpy(hMins.data(), pin + wStats + (candAll ? wStats : 0u), 8); memcpy(hMaxs.data(), pin + wStats + (candAll ? wStats : 0u) + 2u, 8); }
    bandNumValid = (int)hr.stats.numValid;
    bandAllValid = (bandNumValid == (int)nPix);
    haveBits = true;
    if (rq.nBands == 1 && !bandAllValid && bandNumValid > 0 && ctx.auxEvent() && !sendBitsHome()) return false;
    return true;
  };
  if (dByteMask && !buildMask()) return kFailed;
  bool nanSeen = dByteMask ? (hr.stats.hasNaN != 0) : false;
  bool mixedNaN = dByteMask ? (hr.stats.mixedNaN != 0) : false;

  // ---- 2. statistics: per-depth min / max; float: NaN, all-integer, TryRaiseMaxZError candidates
  double maxZErr = rq.maxZErr;
  u32 raiseMask = 0;
  // 8-bit values without a mask, lossless: what the choice between tiling and Huffman coding is made from -- the sizes
  // of the 8 x 8 blocks and the two histograms -- depends on nothing the statistics decide, so both are enqueued right
  // behind the statistics kernel and arrive with the same wait (every wait costs the stream ~50 us of idling)
  struct Speculated
  {
    bool on = false, sizesFresh = false;
    u32* dSizes = nullptr; u32* dOffsets = nullptr; u32* dScratch = nullptr;
    u32* dHisto = nullptr; const u32* dTotal = nullptr;
    u32 total = 0;
    u32 histo[512];
    BandParams bp;
  } spec;
  const bool specWanted = !isFlt && tb == 1 && !dByteMask && !nd.active && rq.maxZErr >= 0 && rq.maxZErr < 1 && rq.version >= 4
    && (nRows > 8 || nCols > 8);
  auto speculate = [&]() -> void
  {
    const int nPos8 = ((nRows + 7) / 8) * ((nCols + 7) / 8);
    spec.dSizes = ctx.allocT<u32>((size_t)nPos8 + 4);
    spec.dOffsets = ctx.allocT<u32>((size_t)nPos8 + 4);
    spec.dScratch = ctx.allocT<u32>((size_t)nPos8 / 1024 + 8);
    if (!spec.dSizes || !spec.dOffsets || !spec.dScratch) return;
    BandParams& b = spec.bp;
    memset(&b, 0, sizeof(b));
    b.nRows = nRows; b.nCols = nCols; b.nDepth = nD; b.dt = dt; b.version = rq.version;
    b.allValid = 1;
    b.maxQ = maxValToQuantize(dt);
    b.maxZErr = 0.5; b.scale = 1; b.invScale = 1;
    b.intLossless = 1;
    b.tryDiff = (rq.version >= 5 && nD > 1) ? 1 : 0;
    b.mb = 8; b.nTV = (nRows + 7) / 8; b.nTH = (nCols + 7) / 8;
    { ProfScope ps(ctx, "tile_sizes"); launchTileSizes(dt, 8, dData, nullptr, b, spec.dSizes, dStatus, st); }
    { ProfScope ps(ctx, "scan_block_sizes"); launchExclusiveScan(spec.dSizes, spec.dOffsets, (u32)nPos8, spec.dScratch, st); }
    enqueueHuffmanHistoDevice(ctx, dt, dData, nullptr, nRows, nCols, nD, spec.dHisto);    // (zeroed by runStats; the counts and the total come home with the statistics)
    spec.dTotal = spec.dOffsets + nPos8;
    spec.on = spec.sizesFresh = true;
  };
  // One kernel sets the statistics kernels' inputs, one gathers their results -- and what was enqueued ahead of the decisions: the
  // two histograms, the blocks' total size -- in pinned memory the host reads after its wait: between the kernels of a band no
  // copy command of a few bytes, and none into pageable memory (each keeps this thread until the stream has reached it).
  auto runStats = [&](int rows, u32 mask) -> bool
  {
    const bool specNow = specWanted && rows == nRows && !spec.on && !haveBits;
    if (specNow && !spec.dHisto) spec.dHisto = ctx.allocT<u32>(512);
    static_assert(sizeof(BandStats) % 4 == 0, "words");
    const u32 wStats = (u32)(sizeof(BandStats) / 4), wKeys = 2u * (u32)nD;
    u32* pin = (u32*)ctx.pinned(((size_t)wStats + 2u * wKeys + 512u + 16u) * 4u);
    if (!pin) return false;
    launchStatsInit(dMins, dMaxs, nD, reinterpret_cast<u32*>(dStats), wStats, (specNow && spec.dHisto) ? spec.dHisto : nullptr, (specNow && spec.dHisto) ? 512u : 0u, st);
    { ProfScope ps(ctx, rows == nRows ? "band_stats" : "band_stats_row0"); launchBandStats(dt, dData, (haveBits && !bandAllValid) ? dNewBits : nullptr, rows, nCols, nD, mask, dMins, dMaxs, dStats, st); }
    if (specNow && spec.dHisto) speculate();
    const u32* const src[5] = { reinterpret_cast<const u32*>(dStats), reinterpret_cast<const u32*>(dMins), reinterpret_cast<const u32*>(dMaxs),
                                spec.on && specNow ? spec.dHisto : nullptr, spec.on && specNow ? spec.dTotal : nullptr };
    const u32 nw[5] = { wStats, wKeys, wKeys, spec.on && specNow ? 512u : 0u, spec.on && specNow ? 1u : 0u };
    launchWordsGather(src, nw, pin, st);
    if (!sync.wait()) return false;
    memcpy(&hr.stats, pin, sizeof(BandStats));
    memcpy(hMins.data(), pin + wStats, (size_t)nD * 8);
    memcpy(hMaxs.data(), pin + wStats + wKeys, (size_t)nD * 8);
    if (spec.on && specNow) { memcpy(spec.histo, pin + wStats + 2u * wKeys, 512 * 4); spec.total = pin[wStats + 2u * wKeys + 512u]; }
    return true;
  };
  if (isFlt && maxZErr > 0)
  {
    // candidates whose error bound beats the request (Lerc2.cpp:1244-1253), pruned on the first row the
    // way the reference prunes after every row (:1277); survivors are then measured over the whole band
    u32 cand = candAll;
    if (cand)
    {
      if (!row0Measured)
      {
        if (!runStats(1, cand)) return kFailed;
        hr.row0 = hr.stats;
      }
      for (int c = 0; c < 9; c++)
        if (((cand >> c) & 1u) && hr.row0.raiseErr[c] / facCand[c] > maxZErr / 2) cand &= ~(1u << c);
      raiseMask = cand;
    }
  }
  if (bandNumValid > 0 && maskStats && raiseMask == 0u && haveBits)
  {
    // (the statistics came with the mask: range, "not all integers", NaN -- and no candidate asks for more)
  }
  else if (bandNumValid > 0)
  {
    if (!runStats(nRows, raiseMask)) return kFailed;
    if (isFlt && hr.stats.hasNaN && !haveBits)
    {
      // NaNs present and no mask yet: derive the mask (NaN in every depth -> invalid) and redo the stats
      if (!buildMask()) return kFailed;
      nanSeen = true;
      mixedNaN = hr.stats.mixedNaN != 0;
      if (bandNumValid > 0 && !runStats(nRows, raiseMask)) return kFailed;
    }
  }
  if (isFlt && nanSeen)
  {
    modifiedMask = true;    // conservative: the reference only flags bands whose mask really changed
    if (mixedNaN && nD > 1)
    {
      if (rq.version >= 6) return kNaN;    // Lerc.cpp:1498-1501 (no noData value to stand in)
      ctx.lastError = "codec < 6 with NaN in some depths of a pixel only (-FLT_MAX stand-ins) is not built";
      return kFailed;
    }
  }
  (void)modifiedMask;

  // ---- mask bookkeeping across bands (Lerc.cpp:717-741)
  std::vector<u8> hBandBits;
  if (haveBits && !bandAllValid && !bitsOnTheWay && !deviceRle)
  {
    const size_t nb = (size_t)((nPix + 7) >> 3);
    {
      hBandBits.resize(nb);
      u8* pin = (u8*)ctx.pinned(nb);    // (a pageable target costs a staging copy at ~1 GB/s)
      if (!pin) return kFailed;
      hipMemcpyAsync(pin, dNewBits, nb, hipMemcpyDeviceToHost, st);
      if (!sync.wait()) return kFailed;
      memcpy(hBandBits.data(), pin, nb);
    }
  }
  if (nanSeen || nd.modifiedMask) anyMaskModified = true;
  bool encMask = (iBand == 0);
  {
    // the reference compares the (filtered) byte masks of consecutive bands; validity bits are equivalent
    const bool compare = (rq.nMasks > 1) || anyMaskModified;    // (an empty vector == all valid)
    if (compare && iBand > 0 && hBandBits != prevByteValid) encMask = true;
    if (rq.nBands > 1 && iBand < rq.nBands - 1) prevByteValid = hBandBits;
  }
  if (encMask)
  {
    ms.allValid = bandAllValid;
    ms.numValid = bandNumValid;
    const size_t nMaskBytes = bitsOnTheWay ? nBitsOnTheWay : deviceRle ? nBitsOnDevice : hBandBits.size();
    ms.hBits = std::move(hBandBits);    // (megabytes for a large raster: moved, not copied)
    if (!bandAllValid)
    {
      if (!ms.dBits) return kFailed;
      hipMemcpyAsync(ms.dBits, dNewBits, nMaskBytes, hipMemcpyDeviceToDevice, st);
    }
  }
  const u8* dBits = ms.allValid ? nullptr : ms.dBits;
  const int numValid = ms.numValid;

  // ---- 3. what Lerc::FilterNoDataAndNaN + Lerc2::ComputeNumBytesNeededToWrite decide from the stats
  Header hd;
  hd.version = rq.version;
  const bool oldCodec = rq.version < 6;    // Lerc::EncodeInternal_v5 (Lerc.cpp:526-624): no all-integer promotion, no noData
  hd.nRows = nRows; hd.nCols = nCols; hd.nDepth = nD; hd.numValid = numValid; hd.dt = dt;
  hd.nBlobsMore = rq.nBands - 1 - iBand;
  std::vector<double> zMinVec(nD, 0), zMaxVec(nD, 0);
  bool allInt = false;
  if (bandNumValid > 0)
  {
    for (int m = 0; m < nD; m++) { zMinVec[m] = statKeyToDouble(dt, hMins[m]); zMaxVec[m] = statKeyToDouble(dt, hMaxs[m]); }
  }
  if (isFlt && !oldCodec)
  {
    if (bandNumValid == 0) maxZErr = 0;    // "tile has no valid data" (Lerc.cpp:1479-1484)
    else
    {
      const double lo = *std::min_element(zMinVec.begin(), zMinVec.end());
      const double hi = *std::max_element(zMaxVec.begin(), zMaxVec.end());
      const double lim = (dt == DT_Float) ? (double)(1L << 23) : (double)((i64)1 << 53);
      allInt = !hr.stats.notAllInt && lo >= -lim && lo <= lim && hi >= -lim && hi <= lim;
      if (allInt) maxZErr = std::max(0.5, floor(maxZErr));
    }
    if (nd.active) { allInt = nd.allInt; maxZErr = nd.maxZErr; }    // decided by the noData filter (it knows the original values)
    hd.isInt = allInt ? 1 : 0;
  }
  else if (nd.active) maxZErr = nd.maxZErr;
  hd.passNoData = nd.needNoData ? 1 : 0;
  hd.noDataVal = nd.needNoData ? nd.noDataNew : 0;
  hd.noDataValOrig = nd.needNoData ? nd.noDataOrig : 0;
  if (maxZErr == 777) maxZErr = -0.01;    // Lerc2.cpp:210-211
  if (!isFlt)
  {
    if (maxZErr < 0)
    {
      // bit plane mode (Lerc2::TryBitPlaneCompression, Lerc2.cpp:1071-1229): drop the low bit planes whose XOR with
      // the neighbours looks like coin flips (|1 - 2 p| < eps); lossless whenever the statistics are inconclusive
      const double eps = -maxZErr;
      maxZErr = 0;
      const int nBits = 8 * dtSize(dt), minCnt = 5000;
      if (bandNumValid >= minCnt)
      {
        u32* dCounts = ctx.allocT<u32>((size_t)nD * 32 + 1);
        if (!dCounts) return kFailed;
        std::vector<u32> hCounts((size_t)nD * 32 + 1);
        { ProfScope ps(ctx, "bitplane_counts"); launchBitPlaneCounts(dt, dData, (haveBits && !bandAllValid) ? dNewBits : nullptr, nRows, nCols, nD, dCounts, st); }
        hipMemcpyAsync(hCounts.data(), dCounts, hCounts.size() * 4, hipMemcpyDeviceToHost, st);
        if (!sync.wait()) return kFailed;
        const double cnt = (double)hCounts[(size_t)nD * 32];
        if (cnt >= minCnt)
        {
          int nCut = 0, lastKept = 0;
          for (int s2 = nBits - 1; s2 >= 0; s2--)
          {
            bool crit = true;
            for (int m = 0; m < nD; m++)
              if (fabs(1 - 2 * ((double)hCounts[(size_t)m * 32 + s2] / cnt)) >= eps) crit = false;
            if (crit && nCut < 2)
            {
              if (nCut == 0) lastKept = s2;
              if (nCut == 1 && s2 < lastKept - 1) { lastKept = s2; nCut = 0; }
              nCut++;
            }
          }
          lastKept = std::max(0, lastKept);
          maxZErr = (double)((1 << lastKept) >> 1);
        }
      }
    }
    maxZErr = std::max(0.5, floor(maxZErr));
  }
  else
  {
    if (maxZErr < 0) return kFailed;
    if (maxZErr > 0 && raiseMask && !allInt && numValid > 0)    // (numValid: Lerc2.cpp:1236)
    {
      for (int c = 0; c < 9; c++)
        if (((raiseMask >> c) & 1u) && hr.stats.raiseErr[c] / facCand[c] <= maxZErr / 2) { maxZErr = errCand[c] / 2; break; }
    }
  }
  TL("statistics read, decisions made");
  hd.maxZErr = maxZErr;
  hd.zMin = hd.zMax = 0;
  hd.mbSize = 8;

  // ---- sections before the pixel data
  const bool needMask = numValid > 0 && numValid < (int)nPix;
  std::vector<u8> rle;
  u32 blobSize = headerBytes(hd.version) + 4;
  bool maskCoded = false;
  auto codeMask = [&]() -> bool    // (called with kernels in flight where there are any: the RLE of 8 MB of bits takes the host a millisecond)
  {
    if (maskCoded) return true;
    maskCoded = true;
    // (the helper must be through in any case: the pinned area takes the blob's prefix next)
    bool helped = rleFuture.valid();
    if (helped) { rle = rleFuture.get(); if (rle.empty()) return false; }
    else if ((bitsOnTheWay || deviceRle) && hipEventSynchronize(ctx.auxEvent()) != hipSuccess) return false;
    if (!(needMask && encMask)) { rle.clear(); return true; }
    if (deviceRle)
    {
      const u32 size = pinRle[0];
      if (size != 0xFFFFFFFFu && size >= 2u && size <= kRleCap)
      {
        if (size > kRleFirst)    // (a mask with many short runs: the rest of its stream)
        {
          if (hipMemcpy((u8*)pinRle + 16 + kRleFirst, dRle + kRleFirst, size - kRleFirst, hipMemcpyDeviceToHost) != hipSuccess) return false;
        }
        rle.assign((const u8*)pinRle + 16, (const u8*)pinRle + 16 + size);
        helped = true;
      }
      else
      {
        // (a mask that hardly compresses: its bits come home after all, and the host codes them)
        ms.hBits.resize(nBitsOnDevice);
        if (hipMemcpy(ms.hBits.data(), dNewBits, nBitsOnDevice, hipMemcpyDeviceToHost) != hipSuccess) return false;
      }
    }
    if (!helped)
    {
      if (bitsOnTheWay) rleEncode(bitsOnTheWay, nBitsOnTheWay, rle);
      else rleEncode(ms.hBits.data(), ms.hBits.size(), rle);
    }
    blobSize += (u32)rle.size();
    return true;
  };

  enum Payload { P_NONE, P_TILING, P_ONESWEEP, P_HUFFMAN, P_FLOAT } payload = P_NONE;
  FplPlan fpl;
  u32 fplPlanes = 0;    // bytes of this band's coded planes, if they were made
  bool writeRanges = false;
  int imageMode = IEM_Tiling;
  HuffmanPlan huff;
  u32 nBytesData = 0;

  BandParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.nRows = nRows; bp.nCols = nCols; bp.nDepth = nD; bp.dt = dt; bp.version = hd.version;
  bp.allValid = ms.allValid ? 1 : 0;
  bp.maxQ = maxValToQuantize(dt);
  bp.maxZErr = maxZErr;
  bp.scale = maxZErr > 0 ? 1 / (2 * maxZErr) : 0;
  bp.invScale = 2 * maxZErr;
  bp.intLossless = (!isFlt && maxZErr == 0.5) ? 1 : 0;
  bp.tryDiff = (hd.version >= 5 && nD > 1 && bp.intLossless) ? 1 : 0;

  u32* dSizes = nullptr;
  u32* dOffsets = nullptr;
  u32* dScratch = nullptr;
  const int nPos8 = ((nRows + 7) / 8) * ((nCols + 7) / 8);
  // A band with a validity mask, one value per pixel, a type of 16 bits or more: its block stream is made by
  // the one-launch encoder's masked form (tile_fast.hip) -- straight into the band's place behind mask and ranges, with the band's
  // FINAL parameters, all decisions about the band having been made above -- instead of tile_sizes + scan + tile_write.  If the band
  // ends up coded another way (16 x 16 blocks, one sweep) that writer comes later in the stream and overwrites it.
  // dStreamed: that stream is in place; a launch that gave up waiting (never seen) leaves the band to the three kernels.
  u8* dStreamed = nullptr;
  u32 nBytesStreamed = 0, streamSums = 0;    // (streamSums: the stream's Fletcher terms, which the kernel collects as it writes)
  bool streamedTried = false;
  auto streamMasked = [&]() -> bool    // false: an error (not: "not applicable")
  {
    if (streamedTried) return true;
    streamedTried = true;
    const bool eligible = maskedStreamingOn() && dBits && !bp.allValid && nD == 1 && hd.version == kCodecVersion
      && dt != DT_Char && dt != DT_Byte && !hd.tryHuffmanInt() && !hd.tryHuffmanFlt() && !nd.active && !bp.tryDiff && fastEncodeOneLaunch() && fastDimsOkRagged(nRows, nCols)
      && ((uintptr_t)dData & 15) == 0 && (!dBandOut || ((uintptr_t)dBandOut & 15) == 0) && (u64)nPix * tb + (u64)nPos8 + 8192 < 0xFFFFFFFFull;
    if (!eligible) return true;
    TL("streamMasked: before codeMask");
    if (!codeMask()) return false;
    TL("streamMasked: mask coded");    // (the mask's length says where the block stream begins; the statistics kernels have covered its coding)
    const u32 nWGt = fastFusedNumWG(dt, nRows, nCols);
    const size_t cellWords = fastFusedCellWords(nWGt), counterWords = fastFusedCounterWords(nWGt);
    const u64 cap = capacityLeft;
    const u32 payloadAt = (u32)(headerBytes(hd.version) + 4 + rle.size() + 2 * (size_t)tb + 1);    // header, mask, ranges, "not one sweep"
    u8* cells = ctx.persistentState(1, cellWords * 8 + 256);
    u8* counters = ctx.persistentState(0, counterWords * 8 + 256);
    u8* dWs = dBandOut;    // (a size query: no payloads, no stores)
    FastEncodeResult* dRes = ctx.allocT<FastEncodeResult>(1);
    FastEncodeResult* hRes = (FastEncodeResult*)ctx.pinned(sizeof(FastEncodeResult));
    if (!cells || !counters || (dBandOut && !dWs) || !dRes || !hRes) return false;
    FastEncodeLaunch fl;
    memset(&fl.fb, 0, sizeof(fl.fb));
    const u32 nG = fastFusedGroups(nWGt), nPG = fastPackGroups(nWGt);
    FastFused& f = fl.fb.fused;
    f.sizeCell = (u64*)cells; f.baseCell = f.sizeCell + nWGt; f.totalCell = f.baseCell + nG; f.raise = f.totalCell + nG;
    f.packPart = (u64*)counters; f.keyPart = f.packPart + nPG + 1;
    f.nWG = nWGt; f.nTiles = 1; f.cellStride = (u32)cellWords; f.counterStride = (u32)counterWords;
    f.tileElems = (u64)nPix; f.outStride = cap;
    f.maskBits = dBits; f.payloadAt = payloadAt;
    f.epoch = ctx.nextEpoch();
    f.publishEpoch = (fastTestGiveUp() & 1u) ? f.epoch ^ 0x5A5A5A5Au : f.epoch;
    f.spinLimit = (fastTestGiveUp() & 1u) ? 8u : (1u << 22);
    fl.fb.result = dRes;
    fl.batch.nTiles = 1; fl.batch.nWG = nWGt; fl.batch.tileElems = (u64)nPix; fl.batch.nBlobsMore = 0;
    BandParams sp = bp;
    sp.mb = 8; sp.nTV = (nRows + 7) / 8; sp.nTH = (nCols + 7) / 8;
    (void)hipGetLastError();
    hipMemsetAsync(dRes, 0, sizeof(FastEncodeResult), st);
    { ProfScope ps(ctx, "masked_encode1"); launchFastEncode(0, sp, maxZErr, 0, dData, dWs, cap, 0, fl.fb, fl.batch, st); }
    if (hipGetLastError() != hipSuccess) { ctx.lastError = "lerc_amd: a streaming encode kernel could not be launched"; return false; }
    hipMemcpyAsync(hRes, dRes, sizeof(FastEncodeResult), hipMemcpyDeviceToHost, st);
    if (!sync.wait()) return false;
    TL("streamMasked: kernel done");
    if (hRes->stuck) { ctx.wipePersistentState(); return true; }    // (the three kernels take the band)
    streamSums = hRes->streamSums;
    nBytesStreamed = hRes->nBytesTiling;    // (a stream that does not fit the buffer was cut off inside it; the size check below says BufferTooSmall)
    dStreamed = dWs ? dWs + payloadAt : reinterpret_cast<u8*>(dRes);    // (size query: only != nullptr counts)
    ctx.lastNote = "masked band: block stream by the one-launch encoder";
    return true;
  };
  auto tilingBytes = [&](int mb, u32& total) -> bool
  {
    bp.mb = mb; bp.nTV = (nRows + mb - 1) / mb; bp.nTH = (nCols + mb - 1) / mb;
    const u32 nPos = (u32)bp.nTV * (u32)bp.nTH;
    if (mb == 8)
    {
      if (!streamMasked()) return false;
      if (dStreamed) { total = nBytesStreamed; return codeMask(); }
    }
    if (mb == 8 && spec.sizesFresh)    // priced behind the statistics already (see above): dSizes / dOffsets hold the result
    {
      spec.sizesFresh = false;
      total = spec.total;
      return codeMask();
    }
    { ProfScope ps(ctx, "tile_sizes"); launchTileSizes(dt, mb, dData, dBits, bp, dSizes, dStatus, st); }
    { ProfScope ps(ctx, "scan_block_sizes"); launchExclusiveScan(dSizes, dOffsets, nPos, dScratch, st); }
    hipMemcpyAsync(&total, dOffsets + nPos, 4, hipMemcpyDeviceToHost, st);
    if (!codeMask()) return false;
    return sync.wait();
  };

  if (numValid > 0)
  {
    hd.zMin = *std::min_element(zMinVec.begin(), zMinVec.end());
    hd.zMax = *std::max_element(zMaxVec.begin(), zMaxVec.end());
    bp.zMaxHdr = hd.zMax;
    bp.checkOverflow = ((dt == DT_Int || dt == DT_UInt) && (hd.zMax - hd.zMin >= 0x7FFFFFFF)) ? 1 : 0;
    if (hd.zMin != hd.zMax)
    {
      writeRanges = hd.version >= 4;    // Lerc2.cpp:260
      if (writeRanges) blobSize += 2u * (u32)nD * (u32)tb;
      const bool constDepths = writeRanges && (0 == memcmp(zMinVec.data(), zMaxVec.data(), nD * sizeof(double)));
      if (!constDepths)
      {
        // (what was enqueued ahead of the decisions only counts if they came out as assumed)
        if (spec.on && !(bp.allValid && bp.intLossless && bp.maxZErr == 0.5 && bp.tryDiff == spec.bp.tryDiff && bp.version == spec.bp.version && !dBits))
          spec.on = spec.sizesFresh = false;
        if (spec.on) { dSizes = spec.dSizes; dOffsets = spec.dOffsets; dScratch = spec.dScratch; }
        else
        {
          dSizes = ctx.allocT<u32>((size_t)nPos8 + 4);
          dOffsets = ctx.allocT<u32>((size_t)nPos8 + 4);
          dScratch = ctx.allocT<u32>((size_t)nPos8 / 1024 + 8);
        }
        if (!dSizes || !dOffsets || !dScratch) return kFailed;

        u32 nBytesTiling = 0;
        if (!tilingBytes(8, nBytesTiling)) return kFailed;
        payload = P_TILING;
        nBytesData = nBytesTiling;
        u32 nBytesHuffman = 0;

        if (hd.tryHuffmanInt())
        {
          TL("before the Huffman plan");
          if (!planHuffman(ctx, dt, dData, dBits, nRows, nCols, nD, hd.version, huff, spec.on ? spec.histo : nullptr)) return kFailed;
          TL("Huffman plan made");
          nBytesHuffman = huff.ok ? huff.nBytes : 0;
          if (huff.ok && nBytesHuffman < nBytesTiling) { payload = P_HUFFMAN; imageMode = huff.imageMode; nBytesData = nBytesHuffman; }
          else huff.ok = false;
        }
        else if (hd.tryHuffmanFlt())
        {
          // lossless float / double: predictor + byte planes + entropy coding, kept if it beats the (raw) blocks by 10 % (Lerc2.cpp:305-328)
          if (!planLosslessFloat(ctx, dt, dData, dByteMask, nd.active, nRows, nCols, nD, fpl)) return kFailed;
          // The reference keeps the coded planes inside its Lerc2 object until a band writes them, and only the
          // nDepth == 1 entry drops planes left over from a band that did not (fpl_Lerc2Ext.cpp:432-452): with
          // nDepth > 1 they count into the next band's length (every later band of a size query, :391-403).
          if (nD == 1) ms.fplStale = 0;
          fplPlanes = fpl.nBytes - 1;
          nBytesHuffman = 1 + ms.fplStale + fplPlanes;
          if ((double)nBytesHuffman < (double)nBytesTiling * 0.9) { payload = P_FLOAT; imageMode = IEM_DeltaDeltaHuffman; nBytesData = nBytesHuffman; }
        }

        const size_t nBytesOneSweep = (size_t)tb * nD * (size_t)numValid;
        // retry with 16 x 16 blocks at low bit rates (Lerc2.cpp:333-357)
        if (((size_t)nBytesTiling * 8 < (size_t)nPix * nD * 1.5)
          && ((size_t)nBytesTiling < 4 * nBytesOneSweep)
          && (nBytesHuffman == 0 || (size_t)nBytesTiling < (size_t)2 * nBytesHuffman)
          && (nRows > 8 || nCols > 8))
        {
          u32 nBytes16 = 0;
          if (!tilingBytes(16, nBytes16)) return kFailed;
          if (nBytes16 <= nBytesData) { nBytesData = nBytes16; payload = P_TILING; imageMode = IEM_Tiling; huff.ok = false; hd.mbSize = 16; }
          else if (payload == P_TILING && !tilingBytes(8, nBytesTiling)) return kFailed;    // restore the 8x8 offsets
          if (hd.mbSize == 8) { bp.mb = 8; bp.nTV = (nRows + 7) / 8; bp.nTH = (nCols + 7) / 8; }
        }
        if (hd.tryHuffmanInt() || hd.tryHuffmanFlt()) nBytesData += 1;
        if (nBytesOneSweep <= (size_t)nBytesData) { payload = P_ONESWEEP; blobSize += 1 + (u32)nBytesOneSweep; }
        else blobSize += 1 + nBytesData;
      }
    }
  }
  if (!codeMask()) return kFailed;
  if ((size_t)blobSize > (size_t)INT_MAX) return kFailed;
  hd.blobSize = (int)blobSize;
  bandBytes = blobSize;
  if (fplPlanes)
  {
    if (dBandOut && payload == P_FLOAT)
    {
      if (ms.fplStale) { ctx.lastError = "lossless float: planes of an earlier band would be written again (reference quirk, not reproduced)"; return kFailed; }
    }
    else ms.fplStale += fplPlanes;
  }
  if (!dBandOut) return kOk;    // size query
  if (blobSize > capacityLeft) return kBufferTooSmall;

  // ---- 4. emit: small sections from the host, pixel payload by kernels
  // (put together in pinned memory -- the mask's RLE can be megabytes, and a pageable source is staged at ~1 GB/s with the
  // stream waiting; the area held the mask bits, which codeMask() has consumed by now)
  const size_t prefixLen = headerBytes(hd.version) + 4 + rle.size() + (writeRanges ? 2 * (size_t)nD * tb : 0) + 2;
  const size_t huffPinAt = (prefixLen + 63) & ~(size_t)63;    // (the Huffman mode's code words and table: emitHuffman)
  const size_t prefixCap = huffPinAt + (payload == P_HUFFMAN ? 2048 + huff.table.size() + 64 : 0);
  u8* prefix = (u8*)ctx.pinnedAux(prefixCap);
  if (!prefix) return kFailed;
  size_t at = 0;
  writeHeader(prefix, hd);
  at = headerBytes(hd.version);
  const int nm = (int)rle.size();
  memcpy(&prefix[at], &nm, 4); at += 4;
  if (nm) { memcpy(&prefix[at], rle.data(), rle.size()); at += rle.size(); }
  if (writeRanges)
  {
    for (int m = 0; m < nD; m++) { const u64 raw = statKeyToRawBits(dt, hMins[m]); putBytes(&prefix[at], raw, tb); at += tb; }
    for (int m = 0; m < nD; m++) { const u64 raw = statKeyToRawBits(dt, hMaxs[m]); putBytes(&prefix[at], raw, tb); at += tb; }
  }
  if (payload != P_NONE)
  {
    prefix[at++] = (payload == P_ONESWEEP) ? 1 : 0;
    if (payload != P_ONESWEEP && (hd.tryHuffmanInt() || hd.tryHuffmanFlt())) prefix[at++] = (u8)imageMode;
  }
  TL("prefix assembled");
  const bool prefixByKernel = payload == P_HUFFMAN && at <= 4096;    // (the Huffman mode's kernel takes a short prefix along: emitHuffman)
  if (!prefixByKernel) hipMemcpyAsync(dBandOut, prefix, at, hipMemcpyHostToDevice, st);
  u8* dPayload = dBandOut + at;

  if (payload == P_TILING && dStreamed && hd.mbSize == 8)    // (in place since streamMasked)
  {
    if (dStreamed != dPayload) { ctx.lastError = "lerc_amd: the masked band's block stream is not where the band's sections end"; return kFailed; }
  }
  else if (payload == P_TILING)
  {
    ProfScope ps(ctx, "tile_write");
    launchTileWrite(dt, hd.mbSize, dData, dBits, bp, dOffsets, dPayload, dStatus, st);
  }
  else if (payload == P_ONESWEEP)
  {
    if (ms.allValid) hipMemcpyAsync(dPayload, dData, (size_t)nElem * tb, hipMemcpyDeviceToDevice, st);
    else
    {
      const i64 nGroups = (nPix + 31) >> 5;
      u32* dCounts = ctx.allocT<u32>((size_t)nGroups + 4);
      u32* dBase = ctx.allocT<u32>((size_t)nGroups + 4);
      u32* dScr = ctx.allocT<u32>((size_t)nGroups / 1024 + 8);
      if (!dCounts || !dBase || !dScr) return kFailed;
      launchMaskGroupCounts(dBits, nPix, dCounts, st);
      launchExclusiveScan(dCounts, dBase, (u32)nGroups, dScr, st);
      launchOneSweep(true, dData, dPayload, dBits, dBase, nPix, nD * tb, st);
    }
  }
  else if (payload == P_FLOAT)
  {
    if (!emitLosslessFloat(ctx, fpl, dPayload)) return kFailed;
  }
  else if (payload == P_HUFFMAN)
  {
    if (!emitHuffman(ctx, dt, dData, dBits, nRows, nCols, nD, huff, dPayload, dStatus, prefix + huffPinAt, prefixByKernel ? dBandOut : nullptr, prefix, (u32)at)) return kFailed;
    TL("Huffman stream enqueued");
  }

  // ---- 5. checksum over blob[14 ..) (Lerc2.cpp:1012-1030), patched into the header (codec 2 has none)
  if (hd.version < 3)
  {
    hipMemcpyAsync(&hr.status, dStatus, sizeof(DeviceStatus), hipMemcpyDeviceToHost, st);
    if (!sync.wait()) return kFailed;
    if (hr.status.error) { ctx.lastError = "device kernel reported an error"; return hr.status.error; }
    if (ctx.profOn()) ctx.profCollect();
    return kOk;
  }
  u64* dFl = ctx.allocT<u64>(kFletcherPartials);
  if (!dFl) return kFailed;
  // (the sums are folded and the header field is written on the device: one wait at the end of the band instead of two)
  if (payload == P_TILING && dStreamed && hd.mbSize == 8 && dStreamed == dPayload && (size_t)(dPayload - dBandOut) + nBytesStreamed == blobSize)
  {
    // the masked band's block stream came with its terms (tile_fast.hip: fusedFlush); what is left to read is what lies in front of it
    ProfScope ps(ctx, "fletcher_enc");
    launchFletcher(dBandOut + 14, (u32)(dPayload - dBandOut) - 14, dFl, st);
    launchFletcherPatchWith(dFl, streamSums, blobSize - 14, dBandOut + 10, st);
  }
  else
  { ProfScope ps(ctx, "fletcher_enc"); launchFletcher(dBandOut + 14, blobSize - 14, dFl, st); launchFletcherPatch(dFl, blobSize - 14, dBandOut + 10, st); }
  hipMemcpyAsync(&hr.status, dStatus, sizeof(DeviceStatus), hipMemcpyDeviceToHost, st);
  if (!sync.wait()) return kFailed;
  TL("band done");
  if (hr.status.error) { ctx.lastError = "device kernel reported an error"; return hr.status.error; }
  if (ctx.profOn()) ctx.profCollect();
  return kOk;
}

static size_t fastEncodeWorkspace(int nRows, int nCols, u32 nTiles)
{
  const size_t nWG = fastEncodeNumWG(nRows, nCols);
  return (size_t)nTiles * (nWG * (kFastBlocksPerWG * sizeof(FastBlockDesc) + 64) + kFastPrefixStage + sizeof(FastEncodeResult)
                           + (fastScanGroups((u32)nWG) + 1) * (4 + 8 * kScanPartWords) + fastPackGroups((u32)nWG) * 16 + fastTicketStride((u32)nWG) * 4 + 256)
    + 65536;
}

static bool prepareFastEncode(Context& ctx, int dt, int nRows, int nCols, double maxZErr, u32 nTiles, u64 tileElems, bool arena,
                              FastEncodeLaunch& fl)
{
  const u32 nWG = fastEncodeNumWG(nRows, nCols);
  const bool isFlt = dt >= DT_Float;
  const size_t nT = nTiles;
  FastEncodeBuffers& fb = fl.fb;
  fl.batch.nTiles = nTiles; fl.batch.nWG = nWG; fl.batch.tileElems = tileElems; fl.batch.nBlobsMore = 0;
  memset(&fb.solo, 0, sizeof(fb.solo));
  memset(&fb.fused, 0, sizeof(fb.fused));
  if (nTiles == 1 && !arena && fastSoloOk(dt, nRows, nCols))
  {
    // one raster: two launches, the pack step's first blocks scan and decide (tile_fast.h)
    memset(&fb, 0, sizeof(fb));
    const bool form = fastEncodeOneLaunch();
    const u32 nWGf = form ? fastFusedNumWG(dt, nRows, nCols) : nWG;    // (the one-launch form counts its own workgroups)
    const size_t nGroups = std::max(fastPackGroups(nWG), fastPackGroups(nWGf)), nFused = fastFusedGroups(std::max(nWG, nWGf));
    // (counters: the pack accumulators, then the one-launch form's key cells; cells: a cell per workgroup, then the
    // one-launch form's group cells and first-row errors)
    u8* counters = ctx.persistentState(0, (nGroups + 1) * 8 + 2 * nGroups * 8 + 256);
    u8* cells = ctx.persistentState(1, ((size_t)std::max(nWG, nWGf) + 2 * nFused + 16) * 8 + 256);
    fb.desc = ctx.allocT<FastBlockDesc>((size_t)nWG * kFastBlocksPerWG);
    fb.wgSize = ctx.allocT<u32>(fastWgStride(nWG) + 4);
    fb.wgMinKey = ctx.allocT<u64>(nWG + 4);
    fb.wgMaxKey = ctx.allocT<u64>(nWG + 4);
    fb.wgFlags = ctx.allocT<u32>(nWG + 4);
    fb.tickets = ctx.allocT<u32>(fastTicketStride(nWG) + 4);
    fb.result = ctx.allocT<FastEncodeResult>(1);
    fb.prefixStage = ctx.allocT<u8>(kFastPrefixStage);
    if (!counters || !cells || !fb.desc || !fb.wgSize || !fb.wgMinKey || !fb.wgMaxKey || !fb.wgFlags || !fb.tickets || !fb.result || !fb.prefixStage)
      return false;
    fb.packPart = (u64*)counters;
    fb.solo.cells = (u64*)cells;
    if (form)
    {
      const size_t nCell = std::max(nWG, nWGf);
      fb.fused.sizeCell = (u64*)cells;
      fb.fused.baseCell = (u64*)cells + nCell;
      fb.fused.totalCell = (u64*)cells + nCell + nFused;
      fb.fused.raise = (u64*)cells + nCell + 2 * nFused;
      fb.fused.packPart = (u64*)counters;
      fb.fused.keyPart = (u64*)counters + nGroups + 1;
      fb.fused.nWG = nWGf;
    }
  }
  else
  {
    fb.desc = ctx.allocT<FastBlockDesc>(nT * nWG * kFastBlocksPerWG);
    fb.wgSize = ctx.allocT<u32>(nT * fastWgStride(nWG) + 4);
    fb.wgBase = ctx.allocT<u32>(nT * fastWgStride(nWG) + 4);
    fb.wgMinKey = ctx.allocT<u64>(nT * nWG + 4);
    fb.wgMaxKey = ctx.allocT<u64>(nT * nWG + 4);
    fb.wgFlags = ctx.allocT<u32>(nT * nWG + 4);
    fb.groupBase = ctx.allocT<u32>(nT * (fastScanGroups(nWG) + 1) + 4);
    fb.scanPart = ctx.allocT<u64>(kScanPartWords * nT * fastScanGroups(nWG) + 4);
    fb.packPart = ctx.allocT<u64>(nT * fastPackGroups(nWG) + 4);
    fb.tickets = ctx.allocT<u32>(nT * fastTicketStride(nWG) + 4);
    fb.result = ctx.allocT<FastEncodeResult>(nT);
    fb.prefixStage = ctx.allocT<u8>(nT * kFastPrefixStage);
    fb.tileOffset = arena ? ctx.allocT<u64>(nT + 1) : nullptr;
    if (!fb.desc || !fb.wgSize || !fb.wgBase || !fb.wgMinKey || !fb.wgMaxKey || !fb.wgFlags || !fb.result
      || !fb.groupBase || !fb.scanPart || !fb.packPart || !fb.tickets
      || !fb.prefixStage || (arena && !fb.tileOffset))
      return false;
  }
  fl.cand = 0;
  if (isFlt)
  {
    static const double errCand[9] = { 1, 0.5, 0.1, 0.05, 0.01, 0.005, 0.001, 0.0005, 0.0001 };
    for (int c = 0; c < 9; c++) if (errCand[c] / 2 > maxZErr) fl.cand |= 1u << c;
  }
  BandParams& bp = fl.bp;
  memset(&bp, 0, sizeof(bp));
  bp.nRows = nRows; bp.nCols = nCols; bp.nDepth = 1; bp.dt = dt; bp.version = kCodecVersion;
  bp.mb = 8; bp.nTV = (nRows + 7) / 8; bp.nTH = (nCols + 7) / 8;
  bp.allValid = 1;
  bp.maxQ = maxValToQuantize(dt);
  bp.maxZErr = isFlt ? maxZErr : std::max(0.5, floor(maxZErr));
  bp.scale = 1 / (2 * bp.maxZErr);
  bp.invScale = 2 * bp.maxZErr;
  bp.intLossless = (!isFlt && bp.maxZErr == 0.5) ? 1 : 0;
  fl.maxZErr = maxZErr;
  return true;
}

static void runFastEncode(Context& ctx, const FastEncodeLaunch& fl, const void* dData, u8* dOut, u64 capacity, u64 arenaBase)
{
  // one kernel per stage (and per profiling group): statistics (+ first-row rounding errors), scan + decide (+ tile
  // placement for batches), pack + checksum
  static const char* kStage[3] = { "fast_stats_sizes", "fast_scan_decide", "fast_pack" };
  const bool ragged = fl.bp.nRows % 8 != 0 || fl.bp.nCols % 8 != 0;
  if (fl.fb.fused.sizeCell && (dOut || ragged))    // one raster, one launch (size queries: only where the two-launch form cannot go)
  {
    ProfScope ps(ctx, "fast_encode1");
    FastEncodeBuffers fb = fl.fb;
    fb.fused.epoch = ctx.nextEpoch();    // (never 0, the tag of a cell nobody has written yet)
    fb.fused.publishEpoch = (fastTestGiveUp() & 1u) ? fb.fused.epoch ^ 0x5A5A5A5Au : fb.fused.epoch;
    fb.fused.spinLimit = (fastTestGiveUp() & 1u) ? 8u : (1u << 22);
    launchFastEncode(0, fl.bp, fl.maxZErr, fl.cand, dData, dOut, capacity, arenaBase, fb, fl.batch, ctx.activeStream());
    return;
  }
  // (no output buffer: the size is known after the decisions -- one raster: the pack step's last workgroup takes them)
  const int nStages = (dOut || fl.fb.solo.cells) ? 3 : 2;
  for (int stage = 0; stage < nStages; stage++)
  {
    if (stage == 1 && fl.fb.solo.cells) continue;    // (one raster: the pack step scans and decides itself)
    ProfScope ps(ctx, kStage[stage]);
    FastEncodeBuffers fb = fl.fb;
    if (stage == 2 && fb.solo.cells) fb.solo.epoch = ctx.nextEpoch();    // (never 0, the tag of a cell nobody has written yet)
    launchFastEncode(stage, fl.bp, fl.maxZErr, fl.cand, dData, dOut, capacity, arenaBase, fb, fl.batch, ctx.activeStream());
  }
}

// single band requests the streaming kernels can take (the same test encodeDevice makes)
static bool encodeStreamingOk(const EncodeRequest& rq)
{
  bool anyNoData = false;
  if (rq.hUsesNoData) for (int i = 0; i < rq.nBands; i++) anyNoData = anyNoData || rq.hUsesNoData[i] != 0;
  return !anyNoData && rq.version == kCodecVersion && rq.maxZErr != 777 && ((uintptr_t)rq.dOut & 15) == 0 && ((uintptr_t)rq.dData & 15) == 0
    && fastEncodeEligible(rq.dt, rq.nRows, rq.nCols, rq.nDepth, rq.nMasks > 0, rq.maxZErr, fastEncodeOneLaunch());
}

bool encodeEnqueueStreaming(Context& ctx, const EncodeRequest& rq, u8* slot)
{
  if (!slot || rq.nBands != 1 || !encodeStreamingOk(rq)) return false;
  ctx.reset();
  if (!ctx.reserve(fastEncodeWorkspace(rq.nRows, rq.nCols, 1) + (1u << 16))) return false;
  FastEncodeLaunch fl;
  if (!prepareFastEncode(ctx, rq.dt, rq.nRows, rq.nCols, rq.maxZErr, 1, 0, false, fl)) return false;
  // (one raster: the deciding block and the last workgroup write the result straight into `slot`, pinned host memory -- no
  // kernel reads it, and a copy kernel behind the encode would cost every call 4 us)
  const bool direct = fl.fb.solo.cells != nullptr;
  if (direct)
  {
    // (wiped first: if a launch fails, what the operation that had the slot before left there must not read as this one's
    // verdict -- "redo" sends the request to the general path, which reports what is wrong)
    FastEncodeResult init;
    memset(&init, 0, sizeof(init));
    init.redo = 1u; init.redoReason = 0x80000000u;
    memcpy(slot, &init, sizeof(init));
    fl.fb.result = reinterpret_cast<FastEncodeResult*>(slot);
  }
  (void)hipGetLastError();
  runFastEncode(ctx, fl, rq.dData, rq.dOut, rq.dOut ? (u64)rq.outCapacity : ~0ull, 0);
  if (hipGetLastError() != hipSuccess) { ctx.lastError = "lerc_amd: a streaming encode kernel could not be launched"; return false; }
  return direct || hipMemcpyAsync(slot, fl.fb.result, sizeof(FastEncodeResult), hipMemcpyDeviceToHost, ctx.activeStream()) == hipSuccess;
}

void encodeStreamingVerdict(Context& ctx, const EncodeRequest& rq, const u8* slot, bool& redo, u32& status, u32& numBytesNeeded, u32& numBytesWritten)
{
  FastEncodeResult hres;
  memcpy(&hres, slot, sizeof(hres));
  if (ctx.profOn()) ctx.profCollect();
  redo = false; status = kOk;
  if (!hres.redo && !hres.stuck)
  {
    ctx.pathCount[0]++;
    numBytesNeeded = hres.blobSize;
    numBytesWritten = rq.dOut ? hres.blobSize : 0;
    return;
  }
  {
    char msg[160];
    snprintf(msg, sizeof(msg), "streaming encode handed the band to the general kernels (redo %u, reason bits 0x%x, stuck %u, blob %u bytes)",
             hres.redo, hres.redoReason, hres.stuck, hres.blobSize);
    ctx.lastNote = msg;
  }
  if (hres.stuck) ctx.wipePersistentState();
  if (rq.dOut && hres.redoReason == 64u && hres.blobSize > rq.outCapacity) { status = kBufferTooSmall; return; }
  redo = true;
}

u32 encodeDevice(Context& ctx, const EncodeRequest& rq, u32& numBytesNeeded, u32& numBytesWritten)
{
  numBytesNeeded = numBytesWritten = 0;
  const int tb = dtSize(rq.dt);
  const i64 nPix = (i64)rq.nRows * rq.nCols;
  const size_t maskBytes = (size_t)((nPix + 7) >> 3) + 64;
  const size_t nPos8 = (size_t)((rq.nRows + 7) / 8) * ((rq.nCols + 7) / 8);
  // workspace: two bit masks, block sizes + offsets + scan scratch, one-sweep ranks, Huffman scratch, small stuff
  size_t need = 2 * maskBytes + 3 * (nPos8 + 1024) * 4 + 3 * ((size_t)(nPix >> 5) + 1024) * 4
    + (rq.dt <= DT_Byte ? huffmanScratchBytes(nPix, rq.nDepth) : 0) + (size_t)rq.nDepth * 16 + (1u << 16);
  bool anyNoData = false;
  if (rq.hUsesNoData) for (int i = 0; i < rq.nBands; i++) anyNoData = anyNoData || rq.hUsesNoData[i] != 0;
  if (anyNoData && !rq.hNoDataValues) return kWrongParam;
  if (anyNoData) need += (size_t)nPix * rq.nDepth * tb + (size_t)nPix + 8192;
  if (rq.nMasks > 0 || rq.dt >= DT_Float) need += maskRleScratchBytes(maskBytes) + (4u << 20) + 8192;    // a large mask is run-length coded on the device
  if (rq.dt >= DT_Float && (rq.maxZErr == 0 || anyNoData) && rq.version >= 6) need += fplEncodeScratchBytes(nPix * rq.nDepth, tb);
  // (a size query, dOut == nullptr, takes the first two steps of the streaming path: statistics and decisions)
  if (rq.version < 2 || rq.version > kCodecVersion) return kWrongParam;
  if (rq.version < 6 && anyNoData) return kWrongParam;    // Lerc.cpp:341-344
  if (rq.version < 4 && rq.nDepth > 1) return kFailed;    // Lerc2::Set refuses (Lerc2.cpp:85-86)
  const bool fastOk = !anyNoData && rq.version == kCodecVersion && rq.maxZErr != 777 && ((uintptr_t)rq.dOut & 15) == 0 && ((uintptr_t)rq.dData & 15) == 0
    && fastEncodeEligible(rq.dt, rq.nRows, rq.nCols, rq.nDepth, rq.nMasks > 0, rq.maxZErr, fastEncodeOneLaunch());
  const u32 nWG = fastOk ? fastEncodeNumWG(rq.nRows, rq.nCols) : 0;
  need += fastOk ? fastEncodeWorkspace(rq.nRows, rq.nCols, 1) : 0;
  const size_t bandCap = (size_t)nPix * tb + 4096;    // a band's blob never exceeds its raw form by more than the small sections
  if (fastOk && rq.nBands > 1 && rq.dOut) need += bandCap + 256;
  (void)nWG;
  if (!ctx.reserve(need)) return kFailed;
  (void)tb;

  // ---- streaming path: everything is decided on the device, one synchronisation at the end.  If an
  // assumption fails (NaN, all-integer floats, raisable error bound, constant image, 16 x 16 retry,
  // raw fallback) the device says so and the general path below redoes the band.
  if (fastOk && rq.nBands > 1)
  {
    // several bands: each is a blob of its own with "bands to follow" in its header (Lerc.cpp:628-789); a band is packed
    // into 16-byte aligned scratch (its place in the output is wherever the band before it ended) and copied over
    hipStream_t st = ctx.activeStream();
    FastEncodeLaunch fl;
    if (!prepareFastEncode(ctx, rq.dt, rq.nRows, rq.nCols, rq.maxZErr, 1, 0, false, fl)) return kFailed;
    u8* dBandBlob = rq.dOut ? ctx.allocT<u8>(bandCap) : nullptr;
    FastEncodeResult* pinRes = (FastEncodeResult*)ctx.pinned(sizeof(FastEncodeResult));
    if (!pinRes || (rq.dOut && !dBandBlob)) return kFailed;
    u64 total = 0;
    bool redo = false;
    for (int iBand = 0; iBand < rq.nBands && !redo; iBand++)
    {
      fl.batch.nBlobsMore = (u32)(rq.nBands - 1 - iBand);
      const u8* dBand = (const u8*)rq.dData + (size_t)iBand * nPix * tb;
      const bool direct = fl.fb.solo.cells != nullptr;    // (the kernels write the result into pinned memory themselves)
      if (direct) { memset(pinRes, 0, sizeof(*pinRes)); pinRes->redo = 1u; pinRes->redoReason = 0x80000000u; fl.fb.result = pinRes; }
      runFastEncode(ctx, fl, dBand, dBandBlob, dBandBlob ? (u64)bandCap : ~0ull, 0);
      if (!direct) hipMemcpyAsync(pinRes, fl.fb.result, sizeof(FastEncodeResult), hipMemcpyDeviceToHost, st);
      if (!ctx.sync()) return kFailed;
      if (pinRes->stuck) ctx.wipePersistentState();
      if (pinRes->redo || pinRes->stuck) { redo = true; break; }
      const u32 bandBytes = pinRes->blobSize;
      if (total + bandBytes > (u64)UINT_MAX) return kDimsTooLarge;
      if (rq.dOut)
      {
        if (total + bandBytes > rq.outCapacity) return kBufferTooSmall;
        hipMemcpyAsync(rq.dOut + total, dBandBlob, bandBytes, hipMemcpyDeviceToDevice, st);
      }
      total += bandBytes;
    }
    if (ctx.profOn()) ctx.profCollect();
    if (!redo)
    {
      if (rq.dOut && !ctx.sync()) return kFailed;
      ctx.pathCount[0]++;
      numBytesNeeded = (u32)total;
      numBytesWritten = rq.dOut ? (u32)total : 0;
      return kOk;
    }
    ctx.reset();
  }
  else if (fastOk)
  {
    FastEncodeResult* pinRes = (FastEncodeResult*)ctx.pinned(sizeof(FastEncodeResult));
    if (!pinRes) return kFailed;
    if (!encodeEnqueueStreaming(ctx, rq, reinterpret_cast<u8*>(pinRes))) return kFailed;
    if (!ctx.sync()) return kFailed;
    bool redo = false;
    u32 status = kOk;
    encodeStreamingVerdict(ctx, rq, reinterpret_cast<const u8*>(pinRes), redo, status, numBytesNeeded, numBytesWritten);
    if (!redo) return status;
    ctx.reset();
  }

  ctx.pathCount[1]++;
  MaskState ms;
  ms.dBits = ctx.allocT<u8>(maskBytes);
  std::vector<u8> prevValid;
  bool anyMaskModified = false;
  u32 total = 0;
  const size_t persistent = maskBytes + 512;    // keep ms.dBits across bands
  for (int iBand = 0; iBand < rq.nBands; iBand++)
  {
    // band scratch is re-used: rewind the bump pointer to just behind the persistent mask
    ctx.reset();
    ctx.alloc(persistent);
    u32 bandBytes = 0;
    u8* dst = rq.dOut ? rq.dOut + total : nullptr;
    const u32 left = rq.dOut ? (rq.outCapacity > total ? rq.outCapacity - total : 0) : 0;
    const u32 rc = encodeBand(ctx, rq, iBand, ms, prevValid, anyMaskModified, dst, left, bandBytes);
    if (rc != kOk) return rc;
    if ((size_t)total + bandBytes > (size_t)UINT_MAX) return kDimsTooLarge;
    total += bandBytes;
  }
  numBytesNeeded = total;
  if (rq.dOut) numBytesWritten = total;
  return kOk;
}

// ------------------------------------------------------------------------------------------------
// A mosaic's worth of independent tiles in one call: every tile becomes its own blob (own header, ranges,
// checksum), byte for byte what encodeDevice() makes of it; the blobs go into one arena at 16-byte aligned offsets.
// Tiles the streaming kernels hand back (constant tiles, NaNs, ...) are encoded one by one behind their sub-batch.
// ------------------------------------------------------------------------------------------------
u32 encodeTilesDevice(Context& ctx, const TilesEncodeRequest& rq, u64& arenaUsed)
{
  arenaUsed = 0;
  if (!rq.dData || !rq.dArena || !rq.hOffsets || !rq.hSizes || rq.nTiles <= 0 || rq.nRows <= 0 || rq.nCols <= 0 || rq.dt < 0 || rq.dt > DT_Double
    || rq.maxZErr < 0 || (rq.slotBytes & 15u) != 0)
    return kWrongParam;
  const bool slotted = rq.slotBytes != 0;    // every tile has its place: no arena to fill front to back
  const int tb = dtSize(rq.dt);
  const u64 tileElems = (u64)rq.nRows * (u64)rq.nCols;
  // (tiles whose sides are no multiples of 8 -- 257 x 257 elevation tiles -- go through the one-launch encoder's ragged form, pixel by
  // pixel where rows do not start on 16-byte boundaries; whole-block tiles must lie 16 bytes apart)
  const bool raggedTile = rq.nRows % 8 != 0 || rq.nCols % 8 != 0;
  const bool fastOk = rq.maxZErr != 777 && ((uintptr_t)rq.dArena & 15) == 0 && ((uintptr_t)rq.dData & 15) == 0
    && (raggedTile ? fastEncodeOneLaunch() : (tileElems * tb) % 16 == 0)
    && fastEncodeEligible(rq.dt, rq.nRows, rq.nCols, 1, false, rq.maxZErr, raggedTile);
  u64 end = 0;    // arena bytes in use

  auto encodeOne = [&](int t) -> u32
  {
    end = slotted ? (u64)t * rq.slotBytes : (end + 15) & ~15ull;
    EncodeRequest one;
    one.dData = (const u8*)rq.dData + (size_t)t * tileElems * tb;
    one.dt = rq.dt; one.nDepth = 1; one.nCols = rq.nCols; one.nRows = rq.nRows; one.nBands = 1; one.nMasks = 0; one.dValidBytes = nullptr;
    one.maxZErr = rq.maxZErr;
    one.dOut = rq.dArena + end;
    one.outCapacity = (u32)std::min<u64>(slotted ? rq.slotBytes : (rq.arenaCapacity > end ? rq.arenaCapacity - end : 0), 0xFFFFFFFFull);
    u32 needed = 0, written = 0;
    const u32 rc = encodeDevice(ctx, one, needed, written);
    if (rc != kOk) return rc;
    rq.hOffsets[t] = end; rq.hSizes[t] = written;
    end += written;
    return kOk;
  };

  if (slotted && rq.arenaCapacity < (u64)rq.nTiles * rq.slotBytes) return kBufferTooSmall;
  if (!fastOk || (slotted && !fastEncodeOneLaunch()))
  {
    for (int t = 0; t < rq.nTiles; t++) { const u32 rc = encodeOne(t); if (rc != kOk) return rc; }
    arenaUsed = slotted ? (u64)rq.nTiles * rq.slotBytes : end;
    return kOk;
  }

  hipStream_t st = ctx.activeStream();
  std::vector<int> redo;
  if (fastEncodeOneLaunch())
  {
    // ---- the one-launch encoder, a tile per blockIdx.y: every tile's blob goes into a slot of its own (half the tile's raw
    // size: a tile that needs more is encoded by itself afterwards), then the tiles are placed and moved into the arena
    const u32 nWGt = fastFusedNumWG(rq.dt, rq.nRows, rq.nCols);
    const size_t cellWords = fastFusedCellWords(nWGt), counterWords = fastFusedCounterWords(nWGt);
    const u64 slotBytes = slotted ? rq.slotBytes : ((tileElems * tb / 2 + 4096) + 15) & ~15ull;
    const bool isFlt = rq.dt >= DT_Float;
    // one launch for tiles [t0, t0 + n), every blob in a slot of slotBytes; redoOut: the tiles it hands back, with the reason bits
    // An arena is filled WITHOUT slots and without a pass that moves the blobs: a tile's last workgroup claims the tile's room with an
    // atomic add on the batch's cursor (tile_fast.h: FastFused::arenaCursor), the tiles lie in the order of their claims.
    // LERC_AMD_TILE_ARENA=copy keeps the slots + k_fast_tile_copy form (a knob for A/B runs and tests).
    // (The emulator runs workgroups one after the other: a workgroup that waits for one BEHIND it waits for ever there.  Emulator builds keep
    // the copy form unless asked, and then give up after a few polls -- which is the hand-back path's test.)
#ifdef HIPSIM
    static const bool cursorMode = []() { const char* e = getenv("LERC_AMD_TILE_ARENA"); return e && strcmp(e, "cursor") == 0; }();
#else
    static const bool cursorMode = []() { const char* e = getenv("LERC_AMD_TILE_ARENA"); return !(e && strcmp(e, "copy") == 0); }();
#endif
    const bool direct = !slotted && cursorMode;
    auto runBatch = [&](int t0, int n, u64 slotBytes, std::vector<std::pair<int, u32> >& redoOut) -> u32
    {
      if (!ctx.reserve((size_t)n * (((slotted || direct) ? 0 : slotBytes) + sizeof(FastEncodeResult) + 8) + (1u << 16))) return kFailed;
      u8* cells = ctx.persistentState(1, (size_t)n * (cellWords + (direct ? 1 : 0)) * 8 + 256);
      u8* counters = ctx.persistentState(0, (size_t)n * counterWords * 8 + 256);
      u8* slots = slotted ? rq.dArena + (size_t)t0 * slotBytes : direct ? rq.dArena : ctx.allocT<u8>((size_t)n * slotBytes);
      // (the results and, behind them, the batch's cursor: cleared by one memset)
      FastEncodeResult* dRes = (FastEncodeResult*)ctx.alloc((size_t)n * sizeof(FastEncodeResult) + 16);
      u64* dCursor = dRes ? reinterpret_cast<u64*>(reinterpret_cast<u8*>(dRes) + (size_t)n * sizeof(FastEncodeResult)) : nullptr;
      static_assert(sizeof(FastEncodeResult) % 8 == 0, "the cursor behind the results is 8-byte aligned");
      u64* dOff = ctx.allocT<u64>((size_t)n + 1);
      if (!cells || !counters || !slots || !dRes || !dOff) return kFailed;
      FastEncodeLaunch fl;
      memset(&fl.fb, 0, sizeof(fl.fb));
      const u32 nG = fastFusedGroups(nWGt), nPG = fastPackGroups(nWGt);
      FastFused& f = fl.fb.fused;
      f.sizeCell = (u64*)cells; f.baseCell = f.sizeCell + nWGt; f.totalCell = f.baseCell + nG; f.raise = f.totalCell + nG;
      f.packPart = (u64*)counters; f.keyPart = f.packPart + nPG + 1;
      f.nWG = nWGt; f.nTiles = (u32)n; f.cellStride = (u32)cellWords; f.counterStride = (u32)counterWords;
      f.tileElems = tileElems; f.outStride = slotBytes;
      end = (end + 15) & ~15ull;
      if (direct)
      {
        f.outStride = 0;
        f.arenaCursor = dCursor;
        f.tileCell = (u64*)cells + (size_t)n * cellWords;    // (epoch-tagged, behind the tiles' own cells)
        f.tileOffset = dOff;
        f.arenaBase = end; f.arenaCapacity = rq.arenaCapacity;
      }
      f.epoch = ctx.nextEpoch();
      f.publishEpoch = (fastTestGiveUp() & 1u) ? f.epoch ^ 0x5A5A5A5Au : f.epoch;
      f.spinLimit = (fastTestGiveUp() & 1u) ? 8u : (1u << 22);
#ifdef HIPSIM
      if (direct) f.spinLimit = 64u;
#endif
      fl.fb.result = dRes;
      fl.batch.nTiles = (u32)n; fl.batch.nWG = nWGt; fl.batch.tileElems = tileElems; fl.batch.nBlobsMore = 0;
      fl.cand = 0;
      if (isFlt)
      {
        static const double errCand[9] = { 1, 0.5, 0.1, 0.05, 0.01, 0.005, 0.001, 0.0005, 0.0001 };
        for (int c = 0; c < 9; c++) if (errCand[c] / 2 > rq.maxZErr) fl.cand |= 1u << c;
      }
      BandParams& bp = fl.bp;
      memset(&bp, 0, sizeof(bp));
      bp.nRows = rq.nRows; bp.nCols = rq.nCols; bp.nDepth = 1; bp.dt = rq.dt; bp.version = kCodecVersion;
      bp.mb = 8; bp.nTV = (rq.nRows + 7) / 8; bp.nTH = (rq.nCols + 7) / 8;
      bp.allValid = 1;
      bp.maxQ = maxValToQuantize(rq.dt);
      bp.maxZErr = isFlt ? rq.maxZErr : std::max(0.5, floor(rq.maxZErr));
      bp.scale = 1 / (2 * bp.maxZErr);
      bp.invScale = 2 * bp.maxZErr;
      bp.intLossless = (!isFlt && bp.maxZErr == 0.5) ? 1 : 0;
      fl.maxZErr = rq.maxZErr;
      (void)hipGetLastError();
      hipMemsetAsync(dRes, 0, (size_t)n * sizeof(FastEncodeResult) + 16, st);    // (the kernels raise `stuck`, nobody else clears it; the cursor)
      {
        ProfScope ps(ctx, "fast_encode1");
        launchFastEncode(0, fl.bp, fl.maxZErr, fl.cand, (const u8*)rq.dData + (size_t)t0 * tileElems * tb, slots, direct ? rq.arenaCapacity : slotBytes, 0, fl.fb, fl.batch, st);
      }
      if (!slotted && !direct)
      {
        ProfScope ps(ctx, "fast_tile_move");
        launchFastTileCopy(dRes, dOff, slots, slotBytes, slotBytes, rq.dArena, (u32)n, end, rq.arenaCapacity, st);
      }
      if (hipGetLastError() != hipSuccess) { ctx.lastError = "lerc_amd: a streaming encode kernel could not be launched"; return kFailed; }
      const size_t resBytes = (size_t)n * sizeof(FastEncodeResult) + 16, offBytes = ((size_t)n + 1) * 8;    // (+ the cursor)
      u8* pin = (u8*)ctx.pinned(resBytes + offBytes);
      if (!pin) return kFailed;
      hipMemcpyAsync(pin, dRes, resBytes, hipMemcpyDeviceToHost, st);
      if (!slotted) hipMemcpyAsync(pin + resBytes, dOff, offBytes, hipMemcpyDeviceToHost, st);
      if (!ctx.sync()) return kFailed;
      if (ctx.profOn()) ctx.profCollect();
      const FastEncodeResult* res = reinterpret_cast<const FastEncodeResult*>(pin);
      const u64* off = reinterpret_cast<const u64*>(pin + resBytes);
      redoOut.clear();
      bool anyStuck = false;
      for (int i = 0; i < n; i++)
      {
        if (res[i].redo || res[i].stuck)
        {
          anyStuck = anyStuck || res[i].stuck != 0;
          if (!slotted && (res[i].redoReason & 128u)) return kBufferTooSmall;    // the arena is full
          redoOut.push_back(std::make_pair(t0 + i, res[i].stuck ? 0u : res[i].redoReason));    // (slotted: a tile that does not fit its slot says so when it is encoded by itself)
          continue;
        }
        rq.hOffsets[t0 + i] = slotted ? (u64)(t0 + i) * slotBytes : off[i];
        rq.hSizes[t0 + i] = res[i].blobSize;
        ctx.pathCount[0]++;
      }
      if (anyStuck) ctx.wipePersistentState();
      if (direct) { u64 claimed; memcpy(&claimed, pin + (size_t)n * sizeof(FastEncodeResult), 8); end += claimed; }    // (incl. the room of tiles that were handed back: holes)
      else if (!slotted) end = off[n];
      return kOk;
    };
    // (a tile that compresses to more than half its raw size -- lossless noise, a small error bound -- does not fit the batch's
    // slots.  A few such tiles are encoded one by one behind the batch; a batch full of them is encoded once more with slots
    // that hold raw blocks, instead of tile after tile with a wait each)
    const u64 slotBig = slotted ? slotBytes : ((tileElems * tb + tileElems / 64 + 4096) + 15) & ~15ull;
    // (a tile is a blockIdx.y: at most 65535 of them per launch)
    const int maxBatch = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)rq.nTiles, 65535), ((size_t)2 << 30) / slotBytes));
    const int maxBig = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)rq.nTiles, 65535), ((size_t)2 << 30) / slotBig));
    std::vector<std::pair<int, u32> > back;
    for (int t0 = 0; t0 < rq.nTiles; t0 += maxBatch)
    {
      const int n = std::min(maxBatch, rq.nTiles - t0);
      const u64 end0 = end;
      const unsigned long long count0 = ctx.pathCount[0];
      u32 rc = runBatch(t0, n, slotBytes, back);
      if (rc != kOk) return rc;
      size_t tooBig = 0;
      for (const auto& r : back) if (r.second == 64u) tooBig++;    // (kRedoCapacity and nothing else)
      if (!slotted && !direct && tooBig > (size_t)std::max(8, n / 32))
      {
        end = end0; ctx.pathCount[0] = count0;
        for (int s0 = t0; s0 < t0 + n; s0 += maxBig)
        {
          rc = runBatch(s0, std::min(maxBig, t0 + n - s0), slotBig, back);
          if (rc != kOk) return rc;
          for (const auto& r : back) { rc = encodeOne(r.first); if (rc != kOk) return rc; }
        }
        continue;
      }
      for (const auto& r : back) { rc = encodeOne(r.first); if (rc != kOk) return rc; }    // (reuses the workspace: the batch is done with it)
    }
    arenaUsed = slotted ? (u64)rq.nTiles * slotBytes : end;
    return kOk;
  }

  // sub-batches keep the workspace bounded (block descriptors are 1/16 of the pixels)
  const size_t perTile = fastEncodeWorkspace(rq.nRows, rq.nCols, 1) - 65536;
  const int maxBatch = (int)std::max<size_t>(1, std::min<size_t>((size_t)rq.nTiles, ((size_t)256 << 20) / perTile));
  for (int t0 = 0; t0 < rq.nTiles; t0 += maxBatch)
  {
    const int n = std::min(maxBatch, rq.nTiles - t0);
    if (!ctx.reserve(fastEncodeWorkspace(rq.nRows, rq.nCols, (u32)n))) return kFailed;
    FastEncodeLaunch fl;
    if (!prepareFastEncode(ctx, rq.dt, rq.nRows, rq.nCols, rq.maxZErr, (u32)n, tileElems, true, fl)) return kFailed;
    end = (end + 15) & ~15ull;
    runFastEncode(ctx, fl, (const u8*)rq.dData + (size_t)t0 * tileElems * tb, rq.dArena, rq.arenaCapacity, end);
    const size_t resBytes = (size_t)n * sizeof(FastEncodeResult), offBytes = ((size_t)n + 1) * 8;
    u8* pin = (u8*)ctx.pinned(resBytes + offBytes);
    if (!pin) return kFailed;
    hipMemcpyAsync(pin, fl.fb.result, resBytes, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(pin + resBytes, fl.fb.tileOffset, offBytes, hipMemcpyDeviceToHost, st);
    if (!ctx.sync()) return kFailed;
    if (ctx.profOn()) ctx.profCollect();
    const FastEncodeResult* res = reinterpret_cast<const FastEncodeResult*>(pin);
    const u64* off = reinterpret_cast<const u64*>(pin + resBytes);
    redo.clear();
    for (int i = 0; i < n; i++)
    {
      if (res[i].redo || res[i].stuck)
      {
        if (res[i].redoReason & 128u) return kBufferTooSmall;    // the arena is full
        redo.push_back(t0 + i);
        continue;
      }
      rq.hOffsets[t0 + i] = off[i];
      rq.hSizes[t0 + i] = res[i].blobSize;
      ctx.pathCount[0]++;
    }
    end = off[n];
    for (int t : redo) { const u32 rc = encodeOne(t); if (rc != kOk) return rc; }    // (reuses the workspace: the batch is done with it)
  }
  arenaUsed = end;
  return kOk;
}

}    // namespace lerc
