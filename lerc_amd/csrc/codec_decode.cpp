// codec_decode.cpp -- lerc_decode() pipeline producing device-resident pixels.
//
// Host logic mirrors Lerc::DecodeTempl (Lerc.cpp:397-521) and Lerc2::Decode (Lerc2.cpp:577-694):
// header + mask + ranges + mode bytes are parsed on the host (tens of bytes; the mask RLE is the
// only sequential piece), everything that touches pixels or the block stream is a HIP kernel.
#include "codec.h"
#include "huffman.h"
#include "fpl.h"
#include "tile_fast.h"
#include <functional>

#include <algorithm>
#include <cstdint>
#include <cstdio>

namespace lerc {

namespace {

// reads small pieces of a blob that lives on the host, the device, or both
struct BlobReader
{
  const u8* h;
  const u8* d;
  u32 n;
  hipStream_t st;
  // bytes already fetched (the head of the current band): served without another device round trip
  const u8* cache = nullptr;
  u64 cacheOff = 0;
  size_t cacheLen = 0;
  Context* ctx = nullptr;    // small device reads go through its pinned mirror (a pageable target costs a staging copy)
  bool read(u64 off, size_t len, u8* dst) const
  {
    if (off + len > n) return false;
    if (h) { memcpy(dst, h + off, len); return true; }
    if (cache && off >= cacheOff && off + len <= cacheOff + cacheLen) { memcpy(dst, cache + (off - cacheOff), len); return true; }
    u8* pin = ctx ? (u8*)ctx->pinned(len < 4096 ? 4096 : len) : nullptr;    // (a pageable target is staged at ~1 GB/s)
    if (hipMemcpyAsync(pin ? pin : dst, d + off, len, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (!(ctx ? ctx->sync() : hipStreamSynchronize(st) == hipSuccess)) return false;
    if (pin) memcpy(dst, pin, len);
    return true;
  }
};

struct BandDesc
{
  u64 offset = 0;
  Header hd;
  size_t hdrLen = 0;
  int numBytesMask = 0;
  u8 head[2048];         // first bytes of the band (header, and for unmasked bands ranges + mode bytes + a Huffman code table)
  size_t headLen = 0;
};

bool readBandHeader(const BlobReader& rd, u64 off, BandDesc& b)
{
  u8* buf = b.head;
  const size_t want = std::min<size_t>(sizeof(b.head), rd.n - off);
  b.headLen = want;
  if (off >= rd.n || !rd.read(off, want, buf)) return false;
  if (!readHeader(buf, want, b.hd, b.hdrLen)) return false;
  if (want < b.hdrLen + 4) return false;
  memcpy(&b.numBytesMask, buf + b.hdrLen, 4);
  if (b.numBytesMask < 0) return false;
  b.offset = off;
  return true;
}

}    // namespace

// ------------------------------------------------------------------------------------------------
// streaming kernels for one band
// ------------------------------------------------------------------------------------------------
// layout of a band's result cell (device, zeroed before the launches, copied back in one piece):
//   [FastDecodeParams, 128 B reserved][fallback bits 16 B][pad to 192]
static const size_t kCellParams = 0, kCellFallback = 128, kCellBytes = 192;
static_assert(sizeof(FastDecodeParams) <= 128, "cell layout");

static bool fastDecodeOneLaunch();    // the one-launch decoders (tile_fast_decode_scan.hip, tile_fast_decode_one.hip); LERC_AMD_DECODE_LAUNCHES=2: discovery + decode as two launches
static int pickForm(Context& ctx, int maxForm, int nRows, int nCols);
static int lowerForm(Context& ctx, int form, int nRows, int nCols) { return form > 1 ? pickForm(ctx, form - 1, nRows, nCols) : 0; }

static size_t fastBandWorkspace(int nRows, int nCols, u32 sizeGiven, u32 nTiles = 1)
{
  const FastWalkPlan wp = makeFastWalkPlan(nRows, nCols, sizeGiven, nTiles);
  const size_t perTile = (size_t)wp.nChunks * (sizeof(FastChunkRec) + (size_t)kDiscWalks * kFastListCap * 2 + 12) + (size_t)wp.nBlocks * 4
    + (size_t)(wp.nChunks / kOneDiscChunks + 1) * 16 + (size_t)(wp.nChunks / kResolveWG + 2) * 4 + 4096 + kResolveWG * sizeof(FastChunkRec);
  return perTile * nTiles + (size_t)kDecodeChunks * kDiscWalks * kFastListCap * 2 + (1u << 16);
}

// Enqueues header check, discovery and decode of nTiles blobs (one band: nTiles == 1, dTileOffset == nullptr);
// nothing is read back here.  dParams [nTiles] and dFallback [4 * nTiles] receive the verdicts; the flags in dFallback
// are raised by writing `epoch` (tile_fast.h), so the cells need no clearing.
// (form 3: the scanning decoder -- one launch, no walks, streams of bit-stuffed blocks; 2: the walking one-launch decoder -- short
// walks over sub-chunks of 1 KiB; 1: discovery + decode as two launches over chunks of 2 KiB, which follow streams the others
// cannot: more tiny blocks in a row, longer stretches without a bit-stuffed block)
static bool launchFastBands(Context& ctx, int form, int dt, int nRows, int nCols, const u8* dBlobs, u32 sizeBound, u32 nTiles, const u64* dTileOffset,
                            const u32* dTileSize, void* dOut, FastDecodeParams* dParams, u32* dFallback, u32 epoch, u8* hCell = nullptr)
{
  hipStream_t st = ctx.activeStream();
  const FastWalkPlan fwp = makeFastWalkPlan(nRows, nCols, sizeBound, nTiles);
  const size_t nT = nTiles, sChunk = fastChunkStride(fwp.nChunks);
  FastDecodeBatch tb;
  tb.nTiles = nTiles; tb.nChunks = fwp.nChunks; tb.nBlocks = fwp.nBlocks; tb.nWaves = fwp.nWaves; tb.discChunks = fwp.discChunks;
  tb.tileElems = (u64)nRows * (u64)nCols; tb.tileOffset = dTileOffset; tb.tileSize = dTileSize;
  FastDecodeBuffers fbuf;
  fbuf.params = dParams;
  fbuf.fallback = dFallback;
  fbuf.hostParams = hCell ? reinterpret_cast<FastDecodeParams*>(hCell + kCellParams) : nullptr;
  fbuf.hostFallback = hCell ? reinterpret_cast<u32*>(hCell + kCellFallback) : nullptr;
  fbuf.epoch = epoch;
  fbuf.publishEpoch = (fastTestGiveUp() & 2u) ? epoch ^ 0x5A5A5A5Au : epoch;
  fbuf.spinLimit = (fastTestGiveUp() & 2u) ? 8u : (1u << 22);
  fbuf.discCell = nullptr;
  fbuf.testRewalk = (fastTestGiveUp() & 4u) ? 1u : 0u;
  fbuf.wgCell = fbuf.wgGroupCell = fbuf.wgAcc = nullptr;
  // The scanning decoder asks for a piece's bytes without waiting for the band header where the blob is expected to reach that far:
  // a batch's sizes are exact; a single band comes with its size (the synchronous calls) or with its buffer's capacity (a decode
  // queued behind the encode that writes the blob) -- then what the context's last band of this shape had, plus an eighth (the
  // bands of one job are alike), else half the raster's raw size.  A guess that is too large costs loads nobody looks at, one
  // that is too small the header's latency in the pieces behind it.
  {
    const u64 raw = (u64)nRows * (u64)nCols * (u64)dtSize(dt);
    u32 spec = sizeBound;
    bool fromHint = false;
    if (!dTileOffset && (u64)sizeBound * 10u >= raw * 9u)    // (as large as the raster itself: a capacity, not a size)
    {
      const bool alike = ctx.scanHint.dt == dt && ctx.scanHint.nRows == nRows && ctx.scanHint.nCols == nCols && ctx.scanHint.end != 0u;
      fromHint = alike;
      const u64 guess = alike ? (u64)ctx.scanHint.end + ctx.scanHint.end / 8u + 65536u : raw / 2u;
      spec = (u32)std::min<u64>(guess, sizeBound);
    }
    fbuf.scanSpecEnd = dTileOffset ? 0xFFFFFFFFu : spec;
    // (... and the launch itself is sized by a guess that comes from a band of this shape: LERC_AMD_SCAN_GRID=0 sizes it by what was given)
    static const bool gridByGuess = []() { const char* e = getenv("LERC_AMD_SCAN_GRID"); return !e || atoi(e) != 0; }();
    fbuf.scanEarly = form == 4 ? 1u : 0u;
    fbuf.scanGridBytes = (gridByGuess && fromHint && spec < sizeBound) ? spec : 0u;
    ctx.lastScanGridBytes = form >= 3 ? (fbuf.scanGridBytes ? fbuf.scanGridBytes : sizeBound) : 0u;
    ctx.lastStreamShape[0] = dt; ctx.lastStreamShape[1] = nRows; ctx.lastStreamShape[2] = nCols;
  }
  fbuf.wgStride = fbuf.wgGroupStride = 0;
  // (epoch-tagged cells, never cleared: they live as long as the context and share its area with the encoder's)
  if (form >= 2)
  {
    // everything in one launch: a cell per workgroup and per group of workgroups, and the groups' checksum accumulators
    // (counters: left zero by the launch's last workgroup)
    const size_t sWg = fastAnyWgStride(sizeBound, dtSize(dt)), sGrp = fastAnyGroupStride(sizeBound, dtSize(dt));
    fbuf.wgStride = (u32)sWg; fbuf.wgGroupStride = (u32)sGrp;
    fbuf.wgCell = (u64*)ctx.persistentState(1, (nT * (sWg + sGrp) + 8) * 8);
    fbuf.wgGroupCell = fbuf.wgCell ? fbuf.wgCell + nT * sWg : nullptr;
    fbuf.wgAcc = (u64*)ctx.persistentState(0, (nT * sGrp + 8) * 8);
    if (!fbuf.wgCell || !fbuf.wgAcc) return false;
    fbuf.recs = nullptr; fbuf.lists = nullptr; fbuf.chunkCell = fbuf.groupCell = fbuf.waveFletcher = nullptr;
    if (form >= 3)
    {
      ProfScope ps(ctx, "fast_decode_scan");
      launchFastDecodeScan(dt, nRows, nCols, tb, dBlobs, sizeBound, fbuf, dOut, st);
      return true;
    }
    ProfScope ps(ctx, "fast_decode_one");
    launchFastDecodeOne(dt, nRows, nCols, tb, dBlobs, sizeBound, fbuf, dOut, st);
    return true;
  }
  fbuf.recs = ctx.allocT<FastChunkRec>(nT * fwp.nChunks + kResolveWG);    // (+ what the resolve step's unconditional loads may touch)
  fbuf.lists = ctx.allocT<u16>((nT * fwp.nChunks + kDecodeChunks) * (size_t)(kDiscWalks * kFastListCap) + 8);    // (+ what the gather step's clamped loads may touch)
  const size_t cellWords = nT * (2 * sChunk + fastGroupStride(fwp.nChunks)) + 8;
  fbuf.chunkCell = (u64*)ctx.persistentState(1, cellWords * 8);
  fbuf.groupCell = fbuf.chunkCell ? fbuf.chunkCell + nT * 2 * sChunk : nullptr;
  fbuf.waveFletcher = ctx.allocT<u64>(2 * nT * (size_t)fwp.nWaves + 4);
  if (!fbuf.recs || !fbuf.lists || !fbuf.chunkCell || !fbuf.waveFletcher)
    return false;
  static const char* kStage[kFastDecodeStages] = { "fast_discover", "fast_decode" };
  for (int stage = 0; stage < kFastDecodeStages; stage++)
  {
    ProfScope ps(ctx, kStage[stage]);
    launchFastDecode(stage, dt, nRows, nCols, tb, dBlobs, sizeBound, fbuf, dOut, st);
  }
  return true;
}

static bool launchFastBand(Context& ctx, int form, int dt, int nRows, int nCols, const u8* dBand, u32 sizeGiven, void* dOutBand, u8* dCell, u32 epoch,
                           u8* hCell = nullptr)
{
  return launchFastBands(ctx, form, dt, nRows, nCols, dBand, sizeGiven, 1, nullptr, nullptr, dOutBand,
                         reinterpret_cast<FastDecodeParams*>(dCell + kCellParams), reinterpret_cast<u32*>(dCell + kCellFallback), epoch, hCell);
}

static bool fastDecodeOneLaunch()
{
  static const bool one = []() { const char* e = getenv("LERC_AMD_DECODE_LAUNCHES"); return !e || atoi(e) != 2; }();
  return one;
}

// The streaming form a request starts with: the highest one at most maxForm that is switched on (LERC_AMD_DECODE_LAUNCHES=2: the
// two-launch form only; LERC_AMD_DECODE_SCAN=0: not the scanning decoder), takes the raster (the scanning decoder: whole 8 x 8
// blocks) and has not just handed a band of this context on: a stream the scanning decoder cannot follow -- many blocks that are not
// bit-stuffed -- costs a launch before the next tier gets it, and the bands of one job are alike, so the next kScanSkip decodes
// start one tier down.
static const u32 kScanSkip = 16;
// Form 4 is form 3 with EARLY counts (tile_fast_decode_scan.hip): a piece says how many blocks it holds as soon as the survivors are
// counted.  A piece whose check or mending comes to another count -- a stream with blocks the scan does not see: constant, all zero,
// raw -- raises a flag, and the band is decoded once more by form 3, which counts late and mends; the bands of one job are alike, so
// the next kScanLate decodes start there.
static const u32 kScanLate = 64;    // (the first time; four times as many after every further one, up to 4 096: Context::scanLateSpan)
static bool scanOffsetsOn()    // (LERC_AMD_SCAN_OFFSETS=0: masked bands keep to the general discovery)
{
  static const bool on = []() { const char* e = getenv("LERC_AMD_SCAN_OFFSETS"); return !e || atoi(e) != 0; }();
  return on;
}
static int pickForm(Context& ctx, int maxForm, int nRows, int nCols)
{
  static const bool scanOn = []() { const char* e = getenv("LERC_AMD_DECODE_SCAN"); return !e || atoi(e) != 0; }();
  static const bool earlyOn = []() { const char* e = getenv("LERC_AMD_SCAN_EARLY"); return !e || atoi(e) != 0; }();
  int f = maxForm > 4 ? 4 : maxForm;
  if (f <= 0) return 0;
  if (!fastDecodeOneLaunch()) return 1;
  if (f >= 3)
  {
    if (!scanOn || !fastDecodeScanEligible(nRows, nCols)) f = 2;
    else if (ctx.scanSkip > 0 && ctx.scanSkipRows == nRows && ctx.scanSkipCols == nCols) { ctx.scanSkip--; f = 2; }    // (bands of the shape that was handed on: the bands of one job are alike, another job's are not)
    else if (f == 4 && (!earlyOn || nRows % 8 != 0 || nCols % 8 != 0)) f = 3;    // (ragged rasters count late: three count values in their filter make false survivors -- which the mending strikes, changing a count -- likelier: one piece in 3 244 of the 8190^2 raster, enough to throw every early launch away)
    else if (f == 4 && ctx.scanLate > 0 && ctx.scanLateRows == nRows && ctx.scanLateCols == nCols) { ctx.scanLate--; f = 3; }    // (the bands of that shape: another job's rasters have streams of their own)
  }
  return f;
}

// reason bits of a tile's / band's four epoch tagged flag cells
static u32 fastFlagBits(const u32* cells, u32 epoch)
{
  u32 bits = 0;
  for (int k = 0; k < 4; k++) if (cells[k] == epoch) bits |= 1u << k;
  return bits;
}

// what the host makes of a band's cell after the sync: 0 = decoded and checksum good, else the reason bits
static u32 fastBandVerdict(const u8* hCell, u32 epoch)
{
  FastDecodeParams hp;
  memcpy(&hp, hCell + kCellParams, sizeof(hp));
  u32 cells[4];
  memcpy(cells, hCell + kCellFallback, 16);
  u32 fb = fastFlagBits(cells, epoch);
  if (!hp.ok) fb |= 0x100u;
  else if (!fb && !hp.checksumOk) fb |= 0x200u;
  return fb;
}

// a band the streaming kernels decoded: its size is the guess for the next band of that shape (launchFastBands)
static void noteBandSize(Context& ctx, const u8* hCell, int dt = -1)
{
  // (the shape from the band's own cell, the type from the request that is being answered: with several decodes of different shapes in
  // flight, the shape the context enqueued LAST is another band's)
  FastDecodeParams hp;
  memcpy(&hp, hCell + kCellParams, sizeof(hp));
  ctx.scanHint.dt = dt >= 0 ? dt : ctx.lastStreamShape[0]; ctx.scanHint.nRows = (int)hp.nRows; ctx.scanHint.nCols = (int)hp.nCols;
  ctx.scanHint.end = hp.blobEnd;
}

// Device-resident single-band blobs: everything is enqueued before a single byte of the blob has been seen by the
// host (the header is checked by k_fast_header); one synchronisation.  handled == false: nothing was decided,
// the caller goes the long way (header read, general kernels, exact status codes).
bool decodeEnqueueStreaming(Context& ctx, const DecodeRequest& rq, u8* slot, u32& epoch)
{
  const int dt = rq.dt, nRows = rq.nRows, nCols = rq.nCols;
  if (!slot || !rq.dBlob || rq.hBlob || rq.nBands != 1 || rq.nDepth != 1 || rq.blobSize < 70 || !rq.dOut) return false;
  if (!fastDecodeEligible(dt, 6, 8, nRows, nCols, 1, true)) return false;
  if (((uintptr_t)rq.dBlob & 15) || ((uintptr_t)rq.dOut & 15)) return false;
  hipStream_t st = ctx.activeStream();
  ctx.reset();
  if (!ctx.reserve(fastBandWorkspace(nRows, nCols, rq.blobSize) + 4096)) return false;
  const size_t cellsBytes = 64 + kCellBytes;
  static_assert(64 + kCellBytes <= Context::kAsyncSlotBytes, "a verdict fits a pinned slot");
  u8* dCells = ctx.allocT<u8>(cellsBytes);
  if (!dCells) return false;
  epoch = ctx.nextEpoch();
  // (the kernels write the verdict -- header check, checksum, flags -- through to `slot`, pinned host memory, as well as to
  // the device cell they read it back from: a copy kernel behind the decode would cost every call 4 us.  The slot is wiped
  // first: if a launch fails, what the operation that had the slot before left there must not read as this one's "ok")
  memset(slot + 64, 0, kCellBytes);
  (void)hipGetLastError();
  const int form = pickForm(ctx, rq.maxForm, nRows, nCols);
  if (form <= 0) return false;
  ctx.lastStreamForm = form;
  if (!launchFastBand(ctx, form, dt, nRows, nCols, rq.dBlob, rq.blobSize, rq.dOut, dCells + 64, epoch, slot + 64)) return false;
  if (hipGetLastError() != hipSuccess) { ctx.lastError = "lerc_amd: a streaming decode kernel could not be launched"; return false; }
  if (rq.nMasks > 0 && rq.dValidBytes) hipMemsetAsync(rq.dValidBytes, 1, (size_t)nRows * nCols, st);    // numValid == nPix or no verdict
  return true;
}

bool decodeStreamingVerdict(Context& ctx, const u8* slot, u32 epoch, u32* bits, int form, int dt)
{
  if (ctx.profOn()) ctx.profCollect();
  if (form < 0) form = ctx.lastStreamForm;
  const u32 verdict = fastBandVerdict(slot + 64, epoch);
  if (bits) *bits = verdict;
  if (verdict & 0x8u) ctx.wipePersistentState();    // (a workgroup gave up waiting: the checksum accumulators may hold residue)
  bool gridTooSmall = false;
  if (form >= 3 && (verdict & 0x4u) && !(verdict & 0x300u))
  {
    // (the launch was sized by the last band of this shape and this one is larger: not the stream's fault -- the guess is forgotten)
    FastDecodeParams hp;
    memcpy(&hp, slot + 64 + kCellParams, sizeof(hp));
    gridTooSmall = ctx.lastScanGridBytes != 0u && fastScanNumWG(hp.blobEnd) > fastScanNumWG(ctx.lastScanGridBytes);
    if (gridTooSmall) ctx.scanHint.end = 0u;
  }
  if (form == 3 && (verdict & 0x7u) && !(verdict & 0x300u) && !gridTooSmall)    // (a stream the scanning decoder does not follow)
  {
    FastDecodeParams hpS;
    memcpy(&hpS, slot + 64 + kCellParams, sizeof(hpS));
    ctx.scanSkip = kScanSkip; ctx.scanSkipRows = (int)hpS.nRows; ctx.scanSkipCols = (int)hpS.nCols;
  }
  if (form == 4 && (verdict & 0x7u) && !(verdict & 0x300u) && !gridTooSmall) {
    FastDecodeParams hpL;
    memcpy(&hpL, slot + 64 + kCellParams, sizeof(hpL));
    ctx.scanLate = ctx.scanLateSpan; ctx.scanLateSpan = std::min<u32>(ctx.scanLateSpan * 4u, 4096u);
    ctx.scanLateRows = (int)hpL.nRows; ctx.scanLateCols = (int)hpL.nCols;
  }    // (an early count was wrong: the late form mends)
  if (verdict)
  {
    char msg[128];
    snprintf(msg, sizeof(msg), "streaming decode (form %d) handed the blob on (reason bits 0x%x)", form, verdict);
    ctx.lastNote = msg;
    if (!(verdict & 0x300u)) ctx.refusalCount[2]++;    // (a tier's launch thrown away; 0x100 / 0x200: the header said "not ours" / the checksum is wrong)
    return false;
  }
  if (form >= 1 && form <= 4) ctx.formCount[std::min(form, 3)]++;
  noteBandSize(ctx, slot + 64, dt);
  return true;
}

// Device-resident single-band blobs: everything is enqueued before a single byte of the blob has been seen by the
// host (the header is checked on the device); one synchronisation.  handled == false: nothing was decided,
// the caller goes the long way (header read, general kernels, exact status codes).
static u32 decodeSpeculative(Context& ctx, const DecodeRequest& rq, bool& handled, bool& tried, u32* bits = nullptr)
{
  handled = false; tried = false;
  u8* pin = (u8*)ctx.pinned(64 + kCellBytes);
  u32 epoch = 0;
  if (!pin || !decodeEnqueueStreaming(ctx, rq, pin, epoch)) return kOk;
  tried = true;
  if (!ctx.sync()) return kFailed;
  handled = decodeStreamingVerdict(ctx, pin, epoch, bits, -1, rq.dt);
  return kOk;
}

// (fastLevel: the streaming form where a band qualifies -- 3 the scanning decoder, 2 the walking one-launch decoder, 1 the two-launch form -- or
// 0: the general kernels only)
static u32 decodeImpl(Context& ctx, const DecodeRequest& rq, int fastLevel, bool& fellBack)
{
  const bool allowFast = fastLevel > 0;
  fellBack = false;
  ctx.lastDecodeStreamed = false;
  hipStream_t st = ctx.activeStream();
  const int dt = rq.dt, nD = rq.nDepth, nCols = rq.nCols, nRows = rq.nRows;
  const int tb = dtSize(dt);
  const i64 nPix = (i64)nRows * nCols;
  const size_t maskBytes = (size_t)((nPix + 7) >> 3);
  BlobReader rd{ rq.hBlob, rq.dBlob, rq.blobSize, st, nullptr, 0, 0, &ctx };

  // ---- walk the band headers (Lerc::GetLercInfo) and check the caller's request against them
  std::vector<BandDesc> bands;
  {
    BandDesc b;
    if (!readBandHeader(rd, 0, b) || b.hd.version < 1)
    {
      u8 magic[10];
      if (rq.blobSize >= 10 && rd.read(0, 10, magic) && memcmp(magic, "CntZImage ", 10) == 0) return decodeLerc1(ctx, rq);    // legacy Lerc1
      return kFailed;    // neither Lerc2 nor Lerc1
    }
    bands.push_back(b);
    u64 total = (u64)b.hd.blobSize;
    if (total > rq.blobSize) return kFailed;
    bool more = (b.hd.version <= 5) || (b.hd.nBlobsMore > 0);
    BandDesc nb;
    while (more && total < rq.blobSize && readBandHeader(rd, total, nb))
    {
      if (nb.hd.nDepth != b.hd.nDepth || nb.hd.nCols != b.hd.nCols || nb.hd.nRows != b.hd.nRows || nb.hd.dt != b.hd.dt) return kFailed;
      if (total + (u64)nb.hd.blobSize > rq.blobSize) return kFailed;
      more = (nb.hd.version <= 5) || (nb.hd.nBlobsMore > 0);
      bands.push_back(nb);
      total += (u64)nb.hd.blobSize;
    }
  }
  int infoMasks = 0, usesNoData = 0;
  for (size_t i = 0; i < bands.size(); i++)
  {
    const BandDesc& b = bands[i];
    if (i == 0) { if (b.numBytesMask > 0 || b.hd.numValid == 0) infoMasks = 1; }
    else if (b.numBytesMask > 0 || b.hd.numValid != bands[0].hd.numValid) infoMasks = 2;
    if (b.hd.passNoData) usesNoData++;
  }
  if (infoMasks > 1) infoMasks = (int)bands.size();
  if (rq.nMasks < infoMasks) return kWrongParam;
  if (rq.nBands > (int)bands.size()) return kWrongParam;
  const bool passNoData = usesNoData && nD > 1;    // Lerc.cpp:430-441: only the _4D entry points can hand the values out
  if (passNoData)
  {
    if (!rq.hUsesNoData || !rq.hNoDataValues) return kHasNoData;
    memset(rq.hUsesNoData, 0, (size_t)rq.nBands);
    memset(rq.hNoDataValues, 0, (size_t)rq.nBands * sizeof(double));
  }

  // ---- workspace
  const u32 nPosMin = (u32)(((nRows + 31) / 32) * ((nCols + 31) / 32));
  (void)nPosMin;
  size_t need = (rq.dBlob ? 0 : (size_t)rq.blobSize + 256) + 2 * (maskBytes + 64) + (1u << 16)
    + (size_t)nD * 8 + 4 * ((size_t)(nPix >> 5) + 1024) * 4;
  // block offsets: one per sub-block for the smallest legal block size we may meet (decided per band below)
  size_t maxSub = 0, maxChunks = 0;
  for (int i = 0; i < rq.nBands; i++)
  {
    const Header& h = bands[i].hd;
    const size_t sub = (size_t)((nRows + h.mbSize - 1) / h.mbSize) * ((nCols + h.mbSize - 1) / h.mbSize) * nD;
    maxSub = std::max(maxSub, sub);
    maxChunks = std::max(maxChunks, (size_t)h.blobSize / 4096 + 2);
  }
  need += maxSub * 4 + maxSub / nD * 2 + 5 * (maxChunks + 1024) * 4 + (dt <= DT_Byte ? huffmanScratchBytes(nPix, nD) : 0);
  {
    size_t cand = 0;    // (k_rank_chunks' table: a word per candidate -- one raw block + 1 of them, 1100 at most, to a chunk of 4 KiB)
    for (int i = 0; i < rq.nBands; i++) cand = std::max(cand, std::min<size_t>(1100, 2 + (size_t)bands[i].hd.mbSize * bands[i].hd.mbSize * tb));
    need += (maxChunks + 2) * (cand + 3) * 4;
  }
  need += fastBandWorkspace(nRows, nCols, rq.blobSize) + 4096;    // streaming path tables
  for (int i = 0; i < rq.nBands; i++)
    if (bands[i].hd.tryHuffmanFlt()) { need += fplDecodeScratchBytes(nPix * nD, tb); break; }
  // masks of some size are decoded where the bits are needed (rle_kernels.hip) instead of on the host between a copy down and a
  // copy up (LERC_AMD_DEVICE_RLE as for the encoder: 0 = never, <bytes> = from masks of that many bytes on; default 256 KB)
  static const size_t kDeviceRleFrom = []() -> size_t { const char* e = getenv("LERC_AMD_DEVICE_RLE"); const long v = e ? atol(e) : 1; return v <= 0 ? ~(size_t)0 : v < 16 ? (size_t)(256u << 10) : (size_t)v; }();
  const size_t kDeviceRleMax = (size_t)4 << 20;    // (16 bytes of tables per byte of the stream)
  size_t rleScratch = 0;
  for (int i = 0; i < rq.nBands; i++)
    if (maskBytes >= kDeviceRleFrom && bands[i].numBytesMask >= 2 && (size_t)bands[i].numBytesMask <= kDeviceRleMax)
      rleScratch = std::max(rleScratch, maskRleDecodeScratchBytes((size_t)bands[i].numBytesMask));
  need += rleScratch + (rleScratch ? 256 : 0);
  if (!ctx.reserve(need)) return kFailed;

  const u8* dBlob = rq.dBlob;
  if (!dBlob)
  {
    u8* stage = ctx.allocT<u8>((size_t)rq.blobSize + 16);
    if (!stage) return kFailed;
    hipMemcpyAsync(stage, rq.hBlob, rq.blobSize, hipMemcpyHostToDevice, st);
    dBlob = stage;
  }
  u8* dBits = ctx.allocT<u8>(maskBytes + 64);
  // everything the host reads back at the end sits together: [status 64 B] then one cell per band (see above);
  // one memset before, one copy after
  const size_t cellsBytes = 64 + (size_t)rq.nBands * kCellBytes;
  u8* dCells = ctx.allocT<u8>(cellsBytes);
  DeviceStatus* dStatus = reinterpret_cast<DeviceStatus*>(dCells);
  double* dZMax = ctx.allocT<double>(nD);
  u64* dFl = ctx.allocT<u64>((size_t)kFletcherPartials * rq.nBands);
  u8* dPixel = ctx.allocT<u8>((size_t)nD * 8);
  if (!dBits || !dCells || !dZMax || !dFl || !dPixel) return kFailed;
  hipMemsetAsync(dCells, 0, cellsBytes, st);

  // host buffers the enqueued copies read from: kept until the call's final synchronisation instead of waiting per band
  std::vector<std::vector<u8> > keepBits;
  bool auxInFlight = false;    // the pinned mask area is the source of a copy that may not have run yet
  bool maskPending = false;    // a band's mask bytes are fetched (maskRle) but not decoded / sent yet: finishMask()
  hipStream_t maskSide = nullptr;    // the mask is being decoded on the device beside the call's stream: finishMask() joins the two
  std::vector<u8> maskRle;
  std::vector<std::vector<double> > keepZMax;
  struct Drain { Context& c; ~Drain() { c.sync(); } } drain{ ctx };    // (destroyed before the buffers above, on every way out)
  bool haveMask = false, maskAllValid = true;
  std::vector<u32> expectChecksum(rq.nBands, 0);
  std::vector<u32> checksumLen(rq.nBands, 0);
  std::vector<u8> small;
  // bands decoded by the streaming kernels: their checksum comes out of the decode kernel itself
  // (a band of its own epoch each: the bands share the context's epoch-tagged cells, and cells left by the band before must
  // not look like this band's)
  struct FastBand { bool used = false; bool offsetsOnly = false; u32 epoch = 0; };    // offsetsOnly: a masked band whose block offsets the scanning decoder's first half found (its flags count, nothing else of the cell)
  std::vector<FastBand> fast(rq.nBands);

  // what a band's kernels take from the workspace (mask tables, chunk tables, block offsets ...) is sized for ONE band: every band
  // starts where the first one did.  The bands' kernels run in the order they are enqueued on the call's stream, and the side
  // stream a mask is decoded on is forked behind everything the band in front enqueued, so a band's tables are dead when the
  // next band's kernels write theirs.
  const size_t bandMark = ctx.used();
  bool maskFromDevice = false;    // this band's mask bits come out of launchMaskRleDecode (the verdict on its stream is still out)
  i64 hostMaskCount = -1;         // set bits among the nPix of a mask the HOST decoded (finishMask), for maskIsSound
  for (int iBand = 0; iBand < rq.nBands; iBand++)
  {
    ctx.rewind(bandMark);
    const BandDesc& bd = bands[iBand];
    const Header& hd = bd.hd;
    if (hd.nDepth != nD || hd.nCols != nCols || hd.nRows != nRows) return kFailed;
    if (hd.dt != dt) { ctx.lastError = "data type of the blob differs from the requested one"; return kFailed; }
    const u8* dBand = dBlob + bd.offset;
    const u32 blobEnd = (u32)hd.blobSize;
    rd.cache = bd.head; rd.cacheOff = bd.offset; rd.cacheLen = bd.headLen;
    u8* dOutBand = (u8*)rq.dOut + (size_t)iBand * nPix * nD * tb;

    // Can the streaming kernels take this band?  (unmasked, nDepth 1, 8 x 8 tiling mode, friendly dimensions;
    // for such bands the ranges and mode bytes sit inside the header bytes we already hold)
    u32 fastDataBegin = 0;
    bool fastBand = false;
    if (allowFast && bd.numBytesMask == 0 && hd.numValid == (int)nPix && hd.zMin != hd.zMax && hd.version >= 3 && hd.maxZErr > 0
      && fastDecodeEligible(dt, hd.version, hd.mbSize, nRows, nCols, nD, true)
      && ((uintptr_t)dBand & 15) == 0 && ((uintptr_t)dOutBand & 15) == 0)
    {
      size_t at0 = bd.hdrLen + 4;
      bool rangesDiffer = true;
      if (hd.version >= 4)
      {
        if (at0 + 2 * (size_t)tb < bd.headLen) rangesDiffer = memcmp(bd.head + at0, bd.head + at0 + tb, tb) != 0;
        at0 += 2 * (size_t)tb;
      }
      if (rangesDiffer && at0 < bd.headLen && bd.head[at0] == 0 && at0 + 1 < (size_t)hd.blobSize)
      {
        fastBand = true;
        fastDataBegin = (u32)(at0 + 1);
      }
    }

    if (hd.version >= 3)
    {
      if (hd.blobSize < 14) return kFailed;
      if (!fastBand) { ProfScope ps(ctx, "fletcher_dec"); launchFletcher(dBand + 14, blobEnd - 14, dFl + (size_t)iBand * kFletcherPartials, st); }
      expectChecksum[iBand] = hd.checksum;
      checksumLen[iBand] = blobEnd - 14;
    }

    // ---- mask (Lerc2::ReadMask, Lerc2.cpp:961-1008)
    u64 at = bd.offset + bd.hdrLen + 4;
    const int nv = hd.numValid;
    if ((nv == 0 || nv == (int)nPix) && bd.numBytesMask != 0) return kFailed;
    if (nv == 0) { haveMask = true; maskAllValid = false; hipMemsetAsync(dBits, 0, maskBytes, st); }
    else if (nv == (int)nPix) { haveMask = true; maskAllValid = true; }
    else if (bd.numBytesMask > 0)
    {
      // fetched now (with the few header bytes behind it, so that the reads further down cost no round trip), decoded
      // and sent to the device by finishMask() -- in tiling mode behind the launch of the chunk walk, which needs no mask
      if (at + (u64)bd.numBytesMask > bd.offset + (u64)blobEnd) return kFailed;
      const size_t extra = std::min<size_t>((size_t)2 * nD * tb + 2, (size_t)(bd.offset + (u64)blobEnd - (at + (u64)bd.numBytesMask)));
      bool onDevice = maskBytes >= kDeviceRleFrom && bd.numBytesMask >= 2 && (size_t)bd.numBytesMask <= kDeviceRleMax;
      u8* scratch = onDevice ? ctx.allocT<u8>(maskRleDecodeScratchBytes((size_t)bd.numBytesMask)) : nullptr;
      if (!scratch) onDevice = false;    // (no room for the tables: the host decodes the stream)
      maskFromDevice = onDevice;
      if (onDevice)
      {
        // decoded on the device, in front of the band's kernels; a damaged stream raises Failed in the status the call ends on.
        // The host fetches the header bytes behind the mask only.
        maskRle.resize(extra);    // (read first: the copy waits for what the stream holds)
        if (extra && !rd.read(at + (u64)bd.numBytesMask, extra, maskRle.data())) return kFailed;
        rd.cache = maskRle.data(); rd.cacheOff = at + (u64)bd.numBytesMask; rd.cacheLen = maskRle.size();
        // (beside the stream: half a dozen small launches and a chain of dependent loads that keep no CU busy, while the
        // stream goes on with the chunk tables, which need no mask)
        maskSide = ctx.auxEvent() ? ctx.forkSide() : nullptr;
        { ProfScope ps(ctx, "mask_rle_decode"); launchMaskRleDecode(dBand + (at - bd.offset), (u32)bd.numBytesMask, dBits, (u32)maskBytes, scratch, dStatus, maskSide ? maskSide : st); }
        haveMask = true; maskAllValid = false;
      }
      else
      {
      maskRle.resize((size_t)bd.numBytesMask + extra);
      if (!rd.read(at, maskRle.size(), maskRle.data())) return kFailed;
      rd.cache = maskRle.data(); rd.cacheOff = at; rd.cacheLen = maskRle.size();
      haveMask = true; maskAllValid = false;
      maskPending = true;
      }
    }
    else if (!haveMask || maskAllValid) return kFailed;    // "use previous mask" without a usable one
    at += (u64)bd.numBytesMask;
    const u8* dMask = maskAllValid ? nullptr : dBits;
    bool wantMaskBytes = iBand < rq.nMasks && rq.dValidBytes;
    auto finishMask = [&](bool joinSide = true) -> bool
    {
      if (maskPending)
      {
        maskPending = false;
        // the bits are put together in pinned memory and travel while the host goes on (the area is free again once its
        // event has passed); without it: a pageable vector that lives until the final sync
        u8* hostBits = nullptr;
        if (ctx.auxEvent())
        {
          if (auxInFlight && hipEventSynchronize(ctx.auxEvent()) != hipSuccess) return false;
          auxInFlight = false;
          hostBits = (u8*)ctx.pinnedAux(maskBytes);
        }
        const bool pinnedBits = hostBits != nullptr;
        if (!pinnedBits) { keepBits.emplace_back(maskBytes, (u8)0); hostBits = keepBits.back().data(); }
        size_t written = 0;
        if (!rleDecode(maskRle.data(), (size_t)bd.numBytesMask, hostBits, maskBytes, &written)) return false;
        if (pinnedBits && written < maskBytes) memset(hostBits + written, 0, maskBytes - written);
        {
          // the valid pixels this mask names (BitMask::CountValidBits over the raster's nPix bits, most significant bit first): what the
          // one-sweep kernel is bounded by -- Lerc2::ReadDataOneSweep asks the mask, not the header (Lerc2.cpp:1379-1385)
          i64 cnt = 0;
          const size_t whole = (size_t)(nPix >> 3);
          size_t i = 0;
          for (; i + 8 <= whole; i += 8) { u64 w8; memcpy(&w8, hostBits + i, 8); cnt += __builtin_popcountll(w8); }
          for (; i < whole; i++) cnt += __builtin_popcount((unsigned)hostBits[i]);
          if (nPix & 7) cnt += __builtin_popcount((unsigned)hostBits[whole] & (0xFF00u >> (nPix & 7)) & 0xFFu);
          hostMaskCount = cnt;
        }
        hipMemcpyAsync(dBits, hostBits, maskBytes, hipMemcpyHostToDevice, st);
        if (pinnedBits) { hipEventRecord(ctx.auxEvent(), st); auxInFlight = true; }
      }
      if (maskSide) ctx.sideInUse();
      if (wantMaskBytes) { wantMaskBytes = false; launchBitsToBytes(dMask, rq.dValidBytes + (size_t)iBand * nPix, nPix, maskSide ? maskSide : st); }
      if (maskSide && joinSide)
      {
        if (hipEventRecord(ctx.auxEvent(), maskSide) != hipSuccess || hipStreamWaitEvent(st, ctx.auxEvent(), 0) != hipSuccess) return false;
        maskSide = nullptr;
      }
      return true;
    };

    // The one-sweep and the Huffman kernels take the mask's word for which pixels the stream holds, and how many: before they
    // run, the mask has to be sound -- out of a run-length stream that was intact (on the device that verdict would otherwise come
    // with the call's last wait, after kernels had gone by a mask that may be anything) -- and its OWN count of valid pixels is what
    // bounds the one-sweep reader (m_bitMask.CountValidBits(), Lerc2.cpp:1379-1385; the header's count is not asked, there or in
    // Lerc2::ReadMask, Lerc2.cpp:961-1008: a blob whose header names another number than its mask holds decodes like the
    // reference's, or fails like it).  A mask the host decoded has been counted by finishMask; one the device decoded costs one
    // wait, for masked bands in those modes only.  (The block kernels of the tiling mode check every block against its valid
    // count themselves.)
    auto maskIsSound = [&](i64& count) -> bool
    {
      count = (i64)nPix;
      if (maskAllValid || !dMask) return true;
      if (!finishMask()) return false;
      if (!maskFromDevice) { count = hostMaskCount >= 0 ? hostMaskCount : (i64)hd.numValid; return true; }
      const i64 nGroups = (nPix + 31) >> 5;
      const size_t mark = ctx.used();
      u32* dCounts = ctx.allocT<u32>((size_t)nGroups + 4);
      u32* dBase = ctx.allocT<u32>((size_t)nGroups + 4);
      u32* dScr = ctx.allocT<u32>((size_t)nGroups / 1024 + 8);
      u32* pinV = (u32*)ctx.pinned(64);
      if (!dCounts || !dBase || !dScr || !pinV) return false;
      launchMaskGroupCounts(dMask, nPix, dCounts, st);
      launchExclusiveScan(dCounts, dBase, (u32)nGroups, dScr, st);
      hipMemcpyAsync(pinV, dBase + nGroups, 4, hipMemcpyDeviceToHost, st);
      hipMemcpyAsync(pinV + 4, dStatus, sizeof(DeviceStatus), hipMemcpyDeviceToHost, st);
      if (!ctx.sync()) return false;
      ctx.rewind(mark);
      const DeviceStatus* hsNow = reinterpret_cast<const DeviceStatus*>(pinV + 4);
      if (hsNow->error) { ctx.lastError = "device kernel reported an error"; return false; }
      count = (i64)pinV[0];
      return true;
    };

    // noData value of this band: handed out, and the remapped value in the decoded pixels turned back into the
    // caller's original one (Lerc.cpp:488-510) -- enqueued when the iteration is left, behind the band's kernels
    struct BandEpilogue
    {
      std::function<void()> fn;
      ~BandEpilogue() { if (fn) fn(); }
    } epilogue;
    if (passNoData)
    {
      rq.hUsesNoData[iBand] = hd.passNoData ? 1 : 0;
      rq.hNoDataValues[iBand] = hd.noDataValOrig;
      if (hd.passNoData && hd.noDataVal != hd.noDataValOrig)
        epilogue.fn = [&, dMask, dOutBand]() { launchNoDataRemap(dt, dOutBand, nullptr, dMask, nPix, nD, hd.noDataVal, hd.noDataValOrig, st); };
    }


    // ---- pixels
    if (nv == 0 && !finishMask()) return kFailed;
    if (nv == 0) { hipMemsetAsync(dOutBand, 0, (size_t)nPix * nD * tb, st); continue; }

    std::vector<double> zMinVec(nD, hd.zMin), zMaxVec(nD, hd.zMax);
    std::vector<u8> pixel((size_t)nD * tb);
    auto fillConst = [&](bool perDepth)
    {
      for (int m = 0; m < nD; m++)
      {
        // (T)hd.zMin resp. (T)m_zMinVec[m] (Lerc2.cpp:2681-2721)
        const u64 bits = typedBits(perDepth ? zMinVec[m] : hd.zMin, dt);
        putBytes(&pixel[(size_t)m * tb], bits, tb);
      }
      hipMemcpyAsync(dPixel, pixel.data(), pixel.size(), hipMemcpyHostToDevice, st);
      launchFill(dOutBand, dPixel, nD * tb, dMask, nPix, st);
      hipStreamSynchronize(st);    // `pixel` dies with this scope
    };
    if (hd.zMin == hd.zMax) { if (!finishMask()) return kFailed; fillConst(false); continue; }

    if (hd.version >= 4)
    {
      small.resize(2 * (size_t)nD * tb);
      if (!rd.read(at, small.size(), small.data())) return kFailed;
      for (int m = 0; m < nD; m++)
      {
        zMinVec[m] = typedFromBits(getBytes(&small[(size_t)m * tb], tb), dt);
        zMaxVec[m] = typedFromBits(getBytes(&small[(size_t)(nD + m) * tb], tb), dt);
      }
      at += small.size();
      if (0 == memcmp(zMinVec.data(), zMaxVec.data(), nD * sizeof(double))) { if (!finishMask()) return kFailed; fillConst(true); continue; }
    }
    u8 flags[2] = { 0, 0 };
    if (at - bd.offset >= blobEnd || !rd.read(at, 1, flags)) return kFailed;
    at += 1;
    if (flags[0])
    {
      if (!finishMask()) return kFailed;
      // one sweep: valid pixels stored raw in order (Lerc2.cpp:1368-1400) -- as many as the MASK names, not as many as the header
      // says: k_one_sweep reads the stream by the mask's ranks, so a header that names fewer pixels than its mask holds in front of a
      // stream cut to match must not get past this bound
      i64 nSweep = 0;
      if (!maskIsSound(nSweep)) return kFailed;
      if ((u64)(at - bd.offset) + (u64)nSweep * nD * tb > blobEnd) return kFailed;
      const u8* src = dBlob + at;
      if (maskAllValid) hipMemcpyAsync(dOutBand, src, (size_t)nPix * nD * tb, hipMemcpyDeviceToDevice, st);
      else
      {
        const i64 nGroups = (nPix + 31) >> 5;
        u32* dCounts = ctx.allocT<u32>((size_t)nGroups + 4);
        u32* dBase = ctx.allocT<u32>((size_t)nGroups + 4);
        u32* dScr = ctx.allocT<u32>((size_t)nGroups / 1024 + 8);
        if (!dCounts || !dBase || !dScr) return kFailed;
        hipMemsetAsync(dOutBand, 0, (size_t)nPix * nD * tb, st);
        launchMaskGroupCounts(dMask, nPix, dCounts, st);
        launchExclusiveScan(dCounts, dBase, (u32)nGroups, dScr, st);
        launchOneSweep(false, src, dOutBand, dMask, dBase, nPix, nD * tb, st);
      }
      continue;
    }
    int imageMode = IEM_Tiling;
    if (hd.tryHuffmanInt() || hd.tryHuffmanFlt())
    {
      if (at - bd.offset >= blobEnd || !rd.read(at, 1, flags + 1)) return kFailed;
      at += 1;
      const int f = flags[1];
      if (f > 3 || (f > 2 && hd.version < 6) || (f > 1 && hd.version < 4)) return kFailed;
      imageMode = f;
    }
    if (imageMode != IEM_Tiling)
    {
      if (!finishMask()) return kFailed;
      if (hd.tryHuffmanFlt())
      {
        if (imageMode != IEM_DeltaDeltaHuffman) return kFailed;    // Lerc2.cpp:674-678
        const u32 rc = decodeLosslessFloat(ctx, dt, rq.hBlob ? rq.hBlob + bd.offset : nullptr, dBand, (u32)(at - bd.offset), blobEnd,
                                           nRows, nCols, nD, dOutBand);
        if (rc != kOk) return rc;
        continue;
      }
      if (!(imageMode == IEM_DeltaHuffman || (hd.version >= 4 && imageMode == IEM_Huffman))) return kFailed;
      { i64 unused = 0; if (!maskIsSound(unused)) return kFailed; }
      const u32 rc = decodeHuffman(ctx, dt, rq.hBlob ? rq.hBlob + bd.offset : nullptr, dBand, (u32)(at - bd.offset), blobEnd,
                                   imageMode, dMask, nRows, nCols, nD, hd.version, dOutBand, dStatus, bd.head, bd.headLen);
      if (rc != kOk) return rc;
      continue;
    }

    // ---- tiling mode: discover the block offsets, then decode
    BandParams bp;
    memset(&bp, 0, sizeof(bp));
    bp.nRows = nRows; bp.nCols = nCols; bp.nDepth = nD; bp.dt = dt; bp.version = hd.version;
    bp.mb = hd.mbSize; bp.nTV = (nRows + bp.mb - 1) / bp.mb; bp.nTH = (nCols + bp.mb - 1) / bp.mb;
    bp.allValid = maskAllValid ? 1 : 0;
    bp.maxQ = maxValToQuantize(dt);
    bp.maxZErr = hd.maxZErr;
    bp.scale = hd.maxZErr > 0 ? 1 / (2 * hd.maxZErr) : 0;
    bp.invScale = 2 * hd.maxZErr;
    bp.zMaxHdr = hd.zMax;

    if (fastBand && fastDataBegin == (u32)(at - bd.offset))
    {
      FastBand& f = fast[iBand];
      f.epoch = ctx.nextEpoch();
      if (!launchFastBand(ctx, fastLevel, dt, nRows, nCols, dBand, blobEnd, dOutBand, dCells + 64 + (size_t)iBand * kCellBytes, f.epoch)) return kFailed;
      ctx.lastDecodeStreamed = true;
      f.used = true;
      if (!finishMask()) return kFailed;    // all valid: the caller's mask bytes become 1s (Lerc.cpp:464-488 always writes them)
      continue;
    }
    if (fastBand)    // launched without its checksum kernel, but did not qualify after all
    {
      ProfScope ps(ctx, "fletcher_dec");
      launchFletcher(dBand + 14, blobEnd - 14, dFl + (size_t)iBand * kFletcherPartials, st);
    }

    keepZMax.push_back(zMaxVec);
    hipMemcpyAsync(dZMax, keepZMax.back().data(), (size_t)nD * 8, hipMemcpyHostToDevice, st);

    DecodeArgs da;
    da.blob = dBand; da.dataBegin = (u32)(at - bd.offset); da.blobEnd = blobEnd;
    da.maskBits = dMask; da.zMaxVec = dZMax; da.out = dOutBand; da.blockOff = nullptr; da.nValidBlk = nullptr;
    WalkPlan wp = makeWalkPlan(bp, da.dataBegin, da.blobEnd, nv);
    wp.test = fastTestGiveUp() & 24u;
    WalkBuffers wb;
    wb.chunkExit = ctx.allocT<u32>(wp.nChunks + 4);
    wb.chunkEntry = ctx.allocT<u32>(wp.nChunks + 4);
    wb.chunkCount = ctx.allocT<u32>(wp.nChunks + 4);
    wb.chunkBase = ctx.allocT<u32>(wp.nChunks + 8);
    wb.blockOff = ctx.allocT<u32>((size_t)wp.nSub + 4);
    wb.scratch = ctx.allocT<u32>(wp.nChunks / 1024 + 8);
    wb.candTab = wp.tabled ? ctx.allocT<u32>((size_t)wp.nChunks * wp.candWindow + 4) : nullptr;
    wb.chunkSub = wp.tabled ? ctx.allocT<u32>(3 * (size_t)wp.nChunks + 4) : nullptr;
    if (wp.tabled && (!wb.candTab || !wb.chunkSub)) return kFailed;
    u16* nValidBlk = nullptr;
    if (wp.uniformN == 0)
    {
      nValidBlk = ctx.allocT<u16>((size_t)bp.nTV * bp.nTH + 4);
      if (!nValidBlk) return kFailed;
    }
    wb.nValidBlk = nValidBlk;
    if (!wb.chunkExit || !wb.chunkEntry || !wb.chunkCount || !wb.chunkBase || !wb.blockOff || !wb.scratch) return kFailed;
    // A band with a mask, 8 x 8 blocks, one value a pixel: the scanning decoder's first half cuts the stream into blocks (tile_fast_decode_scan.hip,
    // MODE 1: count bytes of 1 ... 64) instead of the general discovery below, which looks at every byte position (0.5 ms for the
    // 96 MB of the masked 8192^2 raster).  It hands a stream it does not follow on: the caller repeats the band with level 0.
    const bool scanOffsets = allowFast && !ctx.scanOffsetsBan && nValidBlk && dMask && bp.mb == 8 && nD == 1 && tb >= 2 && hd.version >= 3 && scanOffsetsOn();
    if (scanOffsets)
    {
      const u32 nPos = (u32)bp.nTV * (u32)bp.nTH;
      FastDecodeBuffers fbuf;
      memset(&fbuf, 0, sizeof(fbuf));
      const size_t sWg = fastAnyWgStride(blobEnd, tb), sGrp = fastAnyGroupStride(blobEnd, tb);
      fbuf.wgStride = (u32)sWg; fbuf.wgGroupStride = (u32)sGrp;
      fbuf.wgCell = (u64*)ctx.persistentState(1, (sWg + sGrp + 8) * 8);
      fbuf.wgGroupCell = fbuf.wgCell ? fbuf.wgCell + sWg : nullptr;
      fbuf.wgAcc = (u64*)ctx.persistentState(0, (sGrp + 8) * 8);
      if (!fbuf.wgCell || !fbuf.wgAcc) return kFailed;
      FastBand& f = fast[iBand];
      f.epoch = ctx.nextEpoch();
      f.offsetsOnly = true;
      u8* cell = dCells + 64 + (size_t)iBand * kCellBytes;
      fbuf.params = reinterpret_cast<FastDecodeParams*>(cell + kCellParams);
      fbuf.fallback = reinterpret_cast<u32*>(cell + kCellFallback);
      fbuf.epoch = f.epoch;
      fbuf.publishEpoch = (fastTestGiveUp() & 2u) ? f.epoch ^ 0x5A5A5A5Au : f.epoch;
      fbuf.spinLimit = (fastTestGiveUp() & 2u) ? 8u : (1u << 22);
      // (the scan needs no mask: it runs while the host decodes the mask's RLE and sends the bits)
      { ProfScope ps(ctx, "scan_offsets");
        launchFastScanOffsets(dt, nRows, nCols, dBand, (u32)hd.version, da.dataBegin, blobEnd, wb.blockOff, nPos, fbuf, st); }
      if (!finishMask(false)) return kFailed;
      launchBlockValidCounts(dMask, bp, nValidBlk, maskSide ? maskSide : st);
      if (!finishMask()) return kFailed;    // (joins the side stream, if the mask went that way)
    }
    else
    {
    // the chunk candidates need no mask: they run while the host decodes the mask's RLE and sends the bits
    { ProfScope ps(ctx, "walk_chunks"); launchWalkChunks(bp, wp, da, wb, st); }
    if (!finishMask(false)) return kFailed;
    if (nValidBlk) launchBlockValidCounts(dMask, bp, nValidBlk, maskSide ? maskSide : st);
    if (!finishMask()) return kFailed;    // (joins the side stream, if the mask went that way)
    { ProfScope ps(ctx, "walk_offsets"); launchWalkRest(bp, wp, da, wb, dStatus, st); }
    }
    da.blockOff = wb.blockOff;
    da.nValidBlk = nValidBlk;
    { ProfScope ps(ctx, "tile_decode"); launchTileDecode(dt, bp, da, dStatus, st); }
  }

  // ---- one sync: kernel status + checksums
  bool anyGeneric = false;
  for (int iBand = 0; iBand < rq.nBands; iBand++) if (!fast[iBand].used) anyGeneric = true;
  u8* pin = (u8*)ctx.pinned(cellsBytes);
  if (!pin) return kFailed;
  std::vector<u64> hFl(anyGeneric ? (size_t)kFletcherPartials * rq.nBands : 0);
  hipMemcpyAsync(pin, dCells, cellsBytes, hipMemcpyDeviceToHost, st);
  if (anyGeneric) hipMemcpyAsync(hFl.data(), dFl, hFl.size() * 8, hipMemcpyDeviceToHost, st);
  if (!ctx.sync()) return kFailed;
  const DeviceStatus hs = *reinterpret_cast<const DeviceStatus*>(pin);
  u32 nScanned = 0;
  for (int iBand = 0; iBand < rq.nBands; iBand++)
  {
    if (fast[iBand].offsetsOnly)
    {
      u32 cells[4];
      memcpy(cells, pin + 64 + (size_t)iBand * kCellBytes + kCellFallback, 16);
      const u32 bits = fastFlagBits(cells, fast[iBand].epoch);
      if (bits & 0x8u) ctx.wipePersistentState();
      if (bits)    // the caller repeats with the general kernels' own discovery
      {
        char msg[112];
        snprintf(msg, sizeof(msg), "the scan did not find band %d's blocks (reason bits 0x%x): the general discovery takes it", iBand, bits);
        ctx.lastNote = msg;
        ctx.scanOffsetsBan = true;    // (for the rest of this call)
        ctx.refusalCount[1]++;
        fellBack = true;
        return kOk;
      }
      nScanned++;
      continue;
    }
    if (!fast[iBand].used) continue;
    const u32 verdict = fastBandVerdict(pin + 64 + (size_t)iBand * kCellBytes, fast[iBand].epoch);
    if (verdict & 0x8u) ctx.wipePersistentState();    // (a workgroup gave up waiting: the checksum accumulators may hold residue)
    if (verdict & 0x200u) return kFailed;    // decoded, but the checksum is wrong
    if (fastLevel == 3 && (verdict & 0x7u) && !(verdict & 0x300u)) { ctx.scanSkip = kScanSkip; ctx.scanSkipRows = nRows; ctx.scanSkipCols = nCols; }
    if (fastLevel == 4 && (verdict & 0x7u) && !(verdict & 0x300u)) { ctx.scanLate = ctx.scanLateSpan; ctx.scanLateSpan = std::min<u32>(ctx.scanLateSpan * 4u, 4096u); ctx.scanLateRows = nRows; ctx.scanLateCols = nCols; }
    if (verdict)                             // caller repeats with the general kernels
    {
      char msg[96];
      snprintf(msg, sizeof(msg), "streaming decode handed band %d to the general kernels (reason bits 0x%x)", iBand, verdict);
      ctx.lastNote = msg;
      ctx.refusalCount[2]++;
      fellBack = true;
      return kOk;
    }
    if (fastLevel >= 1 && fastLevel <= 4) ctx.formCount[std::min(fastLevel, 3)]++;
    noteBandSize(ctx, pin + 64 + (size_t)iBand * kCellBytes, dt);
  }
  for (int iBand = 0; iBand < rq.nBands; iBand++)
  {
    if (bands[iBand].hd.version < 3) continue;
    if (fast[iBand].used) continue;    // checked on the device
    u64 A = 0, B = 0;
    for (int i = 0; i < kFletcherPartials; i += 2) { A += hFl[(size_t)iBand * kFletcherPartials + i]; B += hFl[(size_t)iBand * kFletcherPartials + i + 1]; }
    if (fletcherFinish(A, B, checksumLen[iBand]) != expectChecksum[iBand]) return kFailed;
  }
  if (hs.error && nScanned != 0u)
  {
    // The scan's cut of a masked band is a proposal: where it had to guess (a raw block's length is in the mask, not in the stream) the
    // decode kernel, which checks every block against the mask, may refuse it.  The general discovery has the last word.
    ctx.lastNote = "the decode kernels refused the scan's block offsets: the general discovery takes the band";
    ctx.refusalCount[0]++;
    ctx.scanOffsetsBan = true;
    fellBack = true;
    return kOk;
  }
  if (hs.error) { ctx.lastError = "device kernel reported an error"; return hs.error; }
  ctx.formCount[0] += nScanned;    // (lerc_amd_decode_forms: out[0] counts masked bands whose blocks the scan found)
  return kOk;
}

// Host-pointer calls on a device copy of the blob: the same, with the pixels' (and the mask bytes') way back to the host
// enqueued behind the kernels, so that the calling thread waits once.  handled == false: the device did not vouch for
// what it wrote (the caller goes the long way and overwrites it).
u32 decodeSpeculativeToHost(Context& ctx, const DecodeRequest& rq, void* hOut, size_t outBytes, u8* hMask, size_t maskBytes, bool& handled, bool& tried)
{
  handled = false; tried = false;
  u8* pin = (u8*)ctx.pinned(64 + kCellBytes);
  u32 epoch = 0;
  if (!pin || !decodeEnqueueStreaming(ctx, rq, pin, epoch)) return kOk;
  tried = true;
  hipStream_t st = ctx.activeStream();
  hipMemcpyAsync(hOut, rq.dOut, outBytes, hipMemcpyDeviceToHost, st);
  if (hMask && rq.dValidBytes) hipMemcpyAsync(hMask, rq.dValidBytes, maskBytes, hipMemcpyDeviceToHost, st);
  if (!ctx.sync()) return kFailed;
  handled = decodeStreamingVerdict(ctx, pin, epoch, nullptr, -1, rq.dt);
  if (handled) { ctx.pathCount[2]++; ctx.lastDecodeStreamed = true; }
  return kOk;
}

u32 decodeDevice(Context& ctx, const DecodeRequest& rq)
{
  // tiers: the scanning decoder, the walking one-launch decoder, the two-launch form (each follows streams the one in front cannot), the general kernels
  ctx.scanOffsetsBan = false;
  int level = rq.noStreaming ? 0 : pickForm(ctx, rq.maxForm, rq.nRows, rq.nCols);
  // A band whose header says "not for the streaming kernels" (a mask, another mode) costs the blind attempt a launch and a wait before
  // the host reads the header itself.  The bands of one job are alike: after such a refusal the next few requests of that shape go
  // to the header-reading path at once (which still hands an unmasked band to the streaming kernels, a header read later).
  const bool sameShape = ctx.blindShape[0] == rq.dt && ctx.blindShape[1] == rq.nRows && ctx.blindShape[2] == rq.nCols;
  bool blind = true;
  if (ctx.blindSkip > 0 && sameShape) { ctx.blindSkip--; blind = false; }
  while (blind && level > 0)
  {
    bool handled = false, tried = false;
    DecodeRequest r = rq;
    r.maxForm = level;
    u32 bits = 0;
    const u32 src = decodeSpeculative(ctx, r, handled, tried, &bits);
    if (src != kOk) return src;
    if (handled) { ctx.pathCount[2]++; return kOk; }
    if (!tried) break;    // (not a request the streaming kernels take blind: decodeImpl looks at every band)
    if (bits == 0x200u) return kFailed;    // (decoded by the streaming kernels, and the checksum is wrong: no other tier would say anything else)
    if (bits & 0x100u)           // (the header says it is no band for the streaming kernels -- a mask, another mode: the other form would say the same)
    {
      ctx.blindSkip = 8; ctx.blindShape[0] = rq.dt; ctx.blindShape[1] = rq.nRows; ctx.blindShape[2] = rq.nCols;
      break;
    }
    level = lowerForm(ctx, level, rq.nRows, rq.nCols);
  }
  bool fellBack = false;
  u32 rc = decodeImpl(ctx, rq, level, fellBack);
  bool repeated = false;
  while (rc == kOk && fellBack && level > 0)
  {
    level = lowerForm(ctx, level, rq.nRows, rq.nCols);
    repeated = level == 0;
    rc = decodeImpl(ctx, rq, level, fellBack);
  }
  if (rc == kOk) ctx.pathCount[(repeated || !ctx.lastDecodeStreamed) ? 3 : 2]++;
  return rc;
}

// ------------------------------------------------------------------------------------------------
// A mosaic's worth of independent blobs (one per tile, all of one shape and type) in one call.  Blobs the
// streaming kernels refuse (masked, constant, other shapes, damaged ...) are decoded one by one afterwards, which
// also produces the exact status for a bad one.
// ------------------------------------------------------------------------------------------------
u32 decodeTilesDevice(Context& ctx, const TilesDecodeRequest& rq)
{
  if (!rq.dArena || !rq.hOffsets || !rq.hSizes || !rq.dOut || rq.nTiles <= 0 || rq.nRows <= 0 || rq.nCols <= 0 || rq.dt < 0 || rq.dt > DT_Double)
    return kWrongParam;
  const int tbytes = dtSize(rq.dt);
  const u64 tileElems = (u64)rq.nRows * (u64)rq.nCols;
  int batchForm = 0;
  auto decodeOne = [&](int t) -> u32
  {
    DecodeRequest one;
    one.dBlob = rq.dArena + rq.hOffsets[t]; one.blobSize = rq.hSizes[t]; one.dt = rq.dt; one.nDepth = 1; one.nCols = rq.nCols;
    one.nRows = rq.nRows; one.nBands = 1; one.nMasks = 0; one.dValidBytes = nullptr;
    one.dOut = (u8*)rq.dOut + (size_t)t * tileElems * tbytes;
    one.maxForm = batchForm > 0 ? batchForm - 1 : 4;    // the batch's kernels have just been tried: the next form (or, if the batch was the two-launch form, the general kernels); no batch launch: every form
    one.noStreaming = one.maxForm <= 0;
    const u32 rc = decodeDevice(ctx, one);
    if (rc == kFailed)    // (a failed decode leaves zeros, include/lerc_amd.h: the streaming kernels may have written pixels of a damaged blob)
    {
      hipMemsetAsync(one.dOut, 0, (size_t)tileElems * tbytes, ctx.activeStream());
      hipStreamSynchronize(ctx.activeStream());
    }
    return rc;
  };
  bool fastOk = fastDecodeEligible(rq.dt, 6, 8, rq.nRows, rq.nCols, 1, true) && ((uintptr_t)rq.dArena & 15) == 0
    && ((uintptr_t)rq.dOut & 15) == 0 && ((tileElems * tbytes) % 16 == 0 || rq.nRows % 8 != 0 || rq.nCols % 8 != 0);    // (ragged tiles: pixel-wise stores)
  u32 maxSize = 0;
  for (int t = 0; t < rq.nTiles; t++)
  {
    if (rq.hOffsets[t] % 16 != 0 || rq.hSizes[t] < 70) fastOk = false;
    maxSize = std::max(maxSize, rq.hSizes[t]);
  }
  if (!fastOk)
  {
    for (int t = 0; t < rq.nTiles; t++) { const u32 rc = decodeOne(t); if (rc != kOk) return rc; }
    return kOk;
  }

  hipStream_t st = ctx.activeStream();
  const size_t perTile = fastBandWorkspace(rq.nRows, rq.nCols, maxSize, 1) - (1u << 16) + sizeof(FastDecodeParams) + 64;
  const int maxBatch = (int)std::max<size_t>(1, std::min<size_t>((size_t)rq.nTiles, ((size_t)1024 << 20) / perTile));
  std::vector<int> redo;
  for (int t0 = 0; t0 < rq.nTiles; t0 += maxBatch)
  {
    const int n = std::min(maxBatch, rq.nTiles - t0);
    if (!ctx.reserve(fastBandWorkspace(rq.nRows, rq.nCols, maxSize, (u32)n) + (size_t)n * (sizeof(FastDecodeParams) + 64) + 8192)) return kFailed;
    // verdict cells [status 64][params n][fallback 16 n], then the offsets / sizes the kernels index by tile
    const size_t cellsBytes = 64 + (size_t)n * (sizeof(FastDecodeParams) + 16);
    u8* dCells = ctx.allocT<u8>(cellsBytes);
    u64* dOff = ctx.allocT<u64>((size_t)n + 1);
    u32* dSize = ctx.allocT<u32>((size_t)n + 1);
    // (pinned: the tables on their way up, then -- a region of its own, so that nobody has to wait in between -- the verdicts' way back)
    const size_t upBytes = ((size_t)n * 12 + 64 + 63) & ~(size_t)63;
    u8* pinUp = (u8*)ctx.pinned(upBytes + cellsBytes);
    if (!dCells || !dOff || !dSize || !pinUp) return kFailed;
    u8* pin = pinUp + upBytes;
    u64* hOff = reinterpret_cast<u64*>(pinUp);
    u32* hSize = reinterpret_cast<u32*>(pinUp + (size_t)n * 8);
    for (int i = 0; i < n; i++) { hOff[i] = rq.hOffsets[t0 + i]; hSize[i] = rq.hSizes[t0 + i]; }
    hipMemcpyAsync(dOff, hOff, (size_t)n * 8, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(dSize, hSize, (size_t)n * 4, hipMemcpyHostToDevice, st);
    FastDecodeParams* dParams = reinterpret_cast<FastDecodeParams*>(dCells + 64);
    u32* dFallback = reinterpret_cast<u32*>(dCells + 64 + (size_t)n * sizeof(FastDecodeParams));
    const u32 epoch = ctx.nextEpoch();
    // (batches count late, form 3: a tile is a handful of pieces -- nothing to gain from early counts --, and one piece in some thousands
    // has a false survivor that its mending strikes: with early counts that tile would be decoded once more by itself, a wait and
    // a launch of its own; 65 536 tiles: 0.386 of peak against 0.456)
    batchForm = pickForm(ctx, 3, rq.nRows, rq.nCols);
    if (!launchFastBands(ctx, batchForm, rq.dt, rq.nRows, rq.nCols, rq.dArena, maxSize, (u32)n, dOff, dSize,
                         (u8*)rq.dOut + (size_t)t0 * tileElems * tbytes, dParams, dFallback, epoch))
      return kFailed;
    hipMemcpyAsync(pin, dCells, cellsBytes, hipMemcpyDeviceToHost, st);
    if (!ctx.sync()) return kFailed;
    if (ctx.profOn()) ctx.profCollect();
    const FastDecodeParams* hp = reinterpret_cast<const FastDecodeParams*>(pin + 64);
    const u32* hfb = reinterpret_cast<const u32*>(pin + 64 + (size_t)n * sizeof(FastDecodeParams));
    redo.clear();
    bool gaveUp = false;
    for (int i = 0; i < n; i++) gaveUp = gaveUp || (fastFlagBits(hfb + 4 * i, epoch) & 0x8u) != 0u;
    if (gaveUp) ctx.wipePersistentState();    // (a workgroup gave up waiting: the checksum accumulators may hold residue)
    for (int i = 0; i < n; i++)
    {
      const u32 bits = fastFlagBits(hfb + 4 * i, epoch);
      const bool good = hp[i].ok && !bits && hp[i].checksumOk;
      if (good) { ctx.pathCount[2]++; ctx.formCount[std::min(batchForm, 3)]++; }
      else
      {
        if (redo.empty())
        {
          char msg[160];
          snprintf(msg, sizeof(msg), "tile %d of the batch went to the general path (header ok %u, reason bits 0x%x, checksum ok %u)",
                   t0 + i, hp[i].ok, bits, hp[i].checksumOk);
          ctx.lastNote = msg;
        }
        redo.push_back(t0 + i);
      }
    }
    if (batchForm == 3 && redo.size() > (size_t)n / 8) { ctx.scanSkip = kScanSkip; ctx.scanSkipRows = rq.nRows; ctx.scanSkipCols = rq.nCols; }    // (tiles the scanning decoder does not follow: the next batches start one tier down)
    if (batchForm == 4 && redo.size() > (size_t)n / 8) { ctx.scanLate = ctx.scanLateSpan; ctx.scanLateSpan = std::min<u32>(ctx.scanLateSpan * 4u, 4096u); ctx.scanLateRows = rq.nRows; ctx.scanLateCols = rq.nCols; }    // (early counts that were wrong: the next batches count late)
    for (int t : redo) { const u32 rc = decodeOne(t); if (rc != kOk) return rc; }    // (reuses the workspace: the batch is done with it)
  }
  return kOk;
}

}    // namespace lerc
