// lerc1_host.cpp -- host half of the legacy Lerc1 ("CntZImage") decoder: headers, the count part, the band loop.
//
// Reference: CntZImage::read (Lerc1Decode/CntZImage.cpp:74-215), Lerc::DecodeTempl's Lerc1 branch (Lerc.cpp:487-516),
// Lerc::GetLercInfo's (Lerc.cpp:184-266).  Decode only -- the reference has no Lerc1 encoder either.
#include "codec.h"
#include "kernels.h"

#include <cfloat>
#include <cstring>

namespace lerc {

namespace {

const size_t kLerc1Header = 10 + 4 * 4 + 8, kLerc1Part = 3 * 4 + 4;

struct Lerc1Band    // one "CntZImage" of the blob
{
  int width = 0, height = 0;
  double maxZErr = 0;
  bool haveCnt = false;
  int cntTV = 0, cntTH = 0, cntBytes = 0;
  float cntMax = 0;
  u32 cntAt = 0;         // offset of the count part's payload in the blob
  int zTV = 0, zTH = 0, zBytes = 0;
  float zMax = 0;
  u32 zAt = 0;
  u32 end = 0;           // offset behind this band
};

// small pieces of the blob, wherever it lives
struct Fetch
{
  Context& ctx;
  const u8* h;
  const u8* d;
  u32 n;
  bool get(u32 off, size_t len, void* dst) const
  {
    if ((u64)off + len > n) return false;
    if (h) { memcpy(dst, h + off, len); return true; }
    u8* pin = (u8*)ctx.pinned(64);
    if (!pin || len > 64) return false;
    if (hipMemcpyAsync(pin, d + off, len, hipMemcpyDeviceToHost, ctx.activeStream()) != hipSuccess || !ctx.sync()) return false;
    memcpy(dst, pin, len);
    return true;
  }
};

bool readBandLayout(const Fetch& f, u32 at, bool onlyZ, Lerc1Band& b)
{
  u8 hd[kLerc1Header];
  if (!f.get(at, kLerc1Header, hd) || memcmp(hd, "CntZImage ", 10)) return false;
  int version, type;
  memcpy(&version, hd + 10, 4); memcpy(&type, hd + 14, 4); memcpy(&b.height, hd + 18, 4); memcpy(&b.width, hd + 22, 4);
  memcpy(&b.maxZErr, hd + 26, 8);
  if (version != 11 || type != 8) return false;
  if (b.height < 0 || b.width < 0 || b.height > 40000 || b.width > 40000) return false;
  if ((size_t)8 * b.height * b.width > (size_t)INT_MAX) return false;
  if (b.maxZErr > 1e12 || b.maxZErr != b.maxZErr) return false;
  at += (u32)kLerc1Header;
  b.haveCnt = !onlyZ;
  for (int part = onlyZ ? 1 : 0; part < 2; part++)
  {
    u8 ph[kLerc1Part];
    if (!f.get(at, kLerc1Part, ph)) return false;
    int tv, th, nb;
    float mx;
    memcpy(&tv, ph, 4); memcpy(&th, ph + 4, 4); memcpy(&nb, ph + 8, 4); memcpy(&mx, ph + 12, 4);
    at += (u32)kLerc1Part;
    if (nb < 0 || (u64)at + (u32)nb > f.n) return false;
    if (part == 0) { b.cntTV = tv; b.cntTH = th; b.cntBytes = nb; b.cntMax = mx; b.cntAt = at; }
    else { b.zTV = tv; b.zTH = th; b.zBytes = nb; b.zMax = mx; b.zAt = at; }
    at += (u32)nb;
  }
  b.end = at;
  return true;
}

// state of the decoded image that carries over from band to band (the count part is only in the first)
struct Lerc1State
{
  bool allValid = true, noneValid = false;
  u8* dBits = nullptr;
  std::vector<u8> hBits;
  i64 numValid = 0;
};

// decodes one band into dOut (type dt; pixels that are not valid are left alone); the blob is at dBlob
u32 decodeBand(Context& ctx, const Fetch& f, const u8* dBlob, const Lerc1Band& b, Lerc1State& s, int dt, void* dOut)
{
  hipStream_t st = ctx.activeStream();
  const i64 nPix = (i64)b.width * b.height;
  const size_t maskBytes = (size_t)((nPix + 7) >> 3);
  if (b.haveCnt)
  {
    if (b.cntTV != 0 || b.cntTH != 0)
    {
      ctx.lastError = "Lerc1 blob with a tiled (non-binary) count image: not built";
      return kFailed;
    }
    if (b.cntBytes == 0) { s.allValid = b.cntMax > 0; s.noneValid = !s.allValid; s.numValid = s.allValid ? nPix : 0; }
    else
    {
      std::vector<u8> rle((size_t)b.cntBytes);
      s.hBits.assign(maskBytes, 0);
      if (f.h) memcpy(rle.data(), f.h + b.cntAt, rle.size());
      else
      {
        hipMemcpyAsync(rle.data(), dBlob + b.cntAt, rle.size(), hipMemcpyDeviceToHost, st);
        if (!ctx.sync()) return kFailed;
      }
      if (!rleDecode(rle.data(), rle.size(), s.hBits.data(), maskBytes)) return kFailed;
      s.allValid = s.noneValid = false;
      s.numValid = 0;
      for (i64 k = 0; k < nPix; k++) s.numValid += (s.hBits[(size_t)(k >> 3)] & (0x80u >> (k & 7))) ? 1 : 0;
      if (!s.dBits) s.dBits = ctx.allocT<u8>(maskBytes + 64);
      if (!s.dBits) return kFailed;
      hipMemcpyAsync(s.dBits, s.hBits.data(), maskBytes, hipMemcpyHostToDevice, st);
    }
  }
  if (s.noneValid) return kOk;    // nothing to write (the z part is walked by the reference without effect)
  if (b.zTV <= 0 || b.zTH <= 0 || b.zTV > b.height || b.zTH > b.width) return kFailed;    // readTiles, CntZImage.cpp:222-224
  Lerc1Geom g;
  g.width = b.width; g.height = b.height; g.nTV = b.zTV; g.nTH = b.zTH;
  g.tilesAcross = (u32)b.zTH + ((b.width % b.zTH) ? 1u : 0u);
  g.nTiles = g.tilesAcross * ((u32)b.zTV + ((b.height % b.zTV) ? 1u : 0u));
  g.allValid = s.allValid ? 1 : 0;
  g.maxZInImg = b.zMax; g.maxZErr = b.maxZErr;
  const size_t mark = ctx.used();
  u32* dNValid = ctx.allocT<u32>((size_t)g.nTiles + 4);
  u32* dTileOff = ctx.allocT<u32>((size_t)g.nTiles + 4);
  DeviceStatus* dStatus = ctx.allocT<DeviceStatus>(1);
  if (!dNValid || !dTileOff || !dStatus) return kFailed;
  hipMemsetAsync(dStatus, 0, sizeof(DeviceStatus), st);
  const u8* dMask = s.allValid ? nullptr : s.dBits;
  { ProfScope ps(ctx, "lerc1_tile_valid"); launchLerc1TileValid(g, dMask, dNValid, st); }
  { ProfScope ps(ctx, "lerc1_walk"); launchLerc1Walk(g, dBlob + b.zAt, (u32)b.zBytes, dNValid, dTileOff, dStatus, st); }
  { ProfScope ps(ctx, "lerc1_decode"); launchLerc1Decode(dt, g, dBlob + b.zAt, (u32)b.zBytes, dMask, dNValid, dTileOff, dOut, dStatus, st); }
  DeviceStatus* pin = (DeviceStatus*)ctx.pinned(sizeof(DeviceStatus));
  if (!pin) return kFailed;
  hipMemcpyAsync(pin, dStatus, sizeof(DeviceStatus), hipMemcpyDeviceToHost, st);
  if (!ctx.sync()) return kFailed;
  ctx.rewind(mark);
  return pin->error ? kFailed : kOk;
}

}    // namespace

bool isLerc1(const u8* hBlob, u32 n) { return hBlob && n >= 10 && memcmp(hBlob, "CntZImage ", 10) == 0; }

// Lerc::DecodeTempl, Lerc1 branch (Lerc.cpp:487-516)
u32 decodeLerc1(Context& ctx, const DecodeRequest& rq)
{
  hipStream_t st = ctx.activeStream();
  const i64 nPix = (i64)rq.nRows * rq.nCols;
  const int tb = dtSize(rq.dt);
  // (two u32 arrays of one entry per tile; a legal blob may have tiles of a single pixel, so size them for nPix tiles)
  if (!ctx.reserve((rq.dBlob ? 0 : (size_t)rq.blobSize + 256) + (size_t)((nPix + 7) >> 3) + 4096 + 8 * ((size_t)nPix + 4096)))
    return kFailed;
  const u8* dBlob = rq.dBlob;
  if (!dBlob)
  {
    u8* stage = ctx.allocT<u8>((size_t)rq.blobSize + 16);
    if (!stage) return kFailed;
    hipMemcpyAsync(stage, rq.hBlob, rq.blobSize, hipMemcpyHostToDevice, st);
    dBlob = stage;
  }
  const Fetch f{ ctx, rq.hBlob, dBlob, rq.blobSize };
  Lerc1State s;
  u32 at = 0;
  for (int iBand = 0; iBand < rq.nBands; iBand++)
  {
    const size_t hdrBytes = kLerc1Header + (iBand == 0 ? 2 : 1) * kLerc1Part + 1;
    if ((size_t)at + hdrBytes > rq.blobSize) return kFailed;
    Lerc1Band b;
    if (!readBandLayout(f, at, iBand > 0, b)) return kFailed;
    if (b.width != rq.nCols || b.height != rq.nRows) return kFailed;
    u8* dOutBand = (u8*)rq.dOut + (size_t)iBand * nPix * tb;
    const u32 rc = decodeBand(ctx, f, dBlob, b, s, rq.dt, dOutBand);
    if (rc != kOk) return rc;
    if (iBand < rq.nMasks && rq.dValidBytes)
    {
      u8* dm = rq.dValidBytes + (size_t)iBand * nPix;
      if (s.allValid) hipMemsetAsync(dm, 1, (size_t)nPix, st);
      else if (s.noneValid) hipMemsetAsync(dm, 0, (size_t)nPix, st);
      else launchBitsToBytes(s.dBits, dm, nPix, st);
    }
    else if (iBand == 0 && !s.allValid) return kFailed;    // Lerc::Convert: the caller has to take the mask (Lerc.cpp:833-834)
    at = b.end;
  }
  if (!ctx.sync()) return kFailed;
  ctx.pathCount[3]++;
  return kOk;
}

// Lerc::GetLercInfo, Lerc1 branch (Lerc.cpp:184-266): the bands are decoded to find the valid pixel count and the ranges
u32 lerc1BlobInfo(Context& ctx, const u8* hBlob, u32 n, BlobInfo& info, double* mins, double* maxs, size_t nElem)
{
  info = BlobInfo();
  hipStream_t st = ctx.activeStream();
  const size_t hdr0 = kLerc1Header + 2 * kLerc1Part + 1, hdr1 = kLerc1Header + kLerc1Part + 1;
  const Fetch f{ ctx, hBlob, nullptr, n };
  Lerc1Band b0;
  if (hdr0 > n || !readBandLayout(f, 0, false, b0)) return kFailed;
  const i64 nPix = (i64)b0.width * b0.height;
  info.nDepth = 1; info.nCols = b0.width; info.nRows = b0.height; info.dt = DT_Float; info.maxZErr = b0.maxZErr;
  info.zMin = FLT_MAX; info.zMax = -FLT_MAX;
  if (!ctx.reserve((size_t)n + 256 + (size_t)nPix * 4 + (size_t)((nPix + 7) >> 3) + 8 * ((size_t)nPix + 4096) + 65536)) return kFailed;
  u8* dBlob = ctx.allocT<u8>((size_t)n + 16);
  float* dZ = ctx.allocT<float>((size_t)nPix + 4);
  u64* dMin = ctx.allocT<u64>(1);
  u64* dMax = ctx.allocT<u64>(1);
  BandStats* dStats = ctx.allocT<BandStats>(1);
  if (!dBlob || !dZ || !dMin || !dMax || !dStats) return kFailed;
  hipMemcpyAsync(dBlob, hBlob, n, hipMemcpyHostToDevice, st);
  Lerc1State s;
  u32 at = 0;
  bool onlyZ = false;
  while ((size_t)info.blobSize + hdr1 < n)
  {
    Lerc1Band b;
    if (!readBandLayout(f, at, onlyZ, b) || b.width != b0.width || b.height != b0.height
      || decodeBand(ctx, Fetch{ ctx, hBlob, dBlob, n }, dBlob, b, s, DT_Float, dZ) != kOk)
      return info.nBands > 0 ? (u32)kOk : (u32)kFailed;
    onlyZ = true;
    at = b.end;
    info.blobSize = b.end;
    float zMin = FLT_MAX, zMax = -FLT_MAX;
    if (s.numValid > 0)
    {
      const u64 k0 = statKeyInitMin(), k1 = statKeyInitMax();
      u64 hk[2];
      hipMemcpyAsync(dMin, &k0, 8, hipMemcpyHostToDevice, st);
      hipMemcpyAsync(dMax, &k1, 8, hipMemcpyHostToDevice, st);
      hipMemsetAsync(dStats, 0, sizeof(BandStats), st);
      launchBandStats(DT_Float, dZ, s.allValid ? nullptr : s.dBits, b.height, b.width, 1, 0, dMin, dMax, dStats, st);
      hipMemcpyAsync(&hk[0], dMin, 8, hipMemcpyDeviceToHost, st);
      hipMemcpyAsync(&hk[1], dMax, 8, hipMemcpyDeviceToHost, st);
      if (!ctx.sync()) return kFailed;
      zMin = (float)statKeyToDouble(DT_Float, hk[0]); zMax = (float)statKeyToDouble(DT_Float, hk[1]);
    }
    info.numValid = (int)s.numValid;
    info.zMin = std::min(info.zMin, (double)zMin);
    info.zMax = std::max(info.zMax, (double)zMax);
    info.nMasks = s.numValid < nPix ? 1 : 0;
    if (mins && maxs && (size_t)info.nBands < nElem) { mins[info.nBands] = zMin; maxs[info.nBands] = zMax; }
    info.nBands++;
  }
  return kOk;
}

}    // namespace lerc
