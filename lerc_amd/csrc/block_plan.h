// block_plan.h -- per-block encoding decision shared by the general and the streaming encoder kernels.
#pragma once
#include "lerc_common.h"

namespace lerc {

struct Plan
{
  int nBytes;
  int kind;       // 0 all-zero (flag 2) | 1 raw (flag 0) | 2 constant (flag 3) | 3 bit-stuffed simple | 4 bit-stuffed LUT
  int tc, dtRed;
  u32 maxElem;
};

// Lerc2::NumBytesTile, Lerc2.h:416-453.  mv = (zMax - zMin) * scale (only meaningful when maxZErr > 0),
// qMax / nDistinct are only meaningful when tryLut && the block quantises.
template<class Z>
__device__ __forceinline__ Plan planBlock(const BandParams& p, int n, Z zMin, Z zMax, int dtZ, bool tryLut, double mv,
                                          u32 qMax, u32 nDistinct)
{
  Plan pl;
  pl.tc = 0; pl.dtRed = dtZ; pl.maxElem = 0;
  if (n == 0 || (zMin == 0 && zMax == 0)) { pl.nBytes = 1; pl.kind = 0; return pl; }
  const int raw = 1 + n * (int)sizeof(Z);
  const double e = p.maxZErr;
  if ((e == 0 && zMax > zMin) || (e > 0 && mv > (double)p.maxQ)) { pl.nBytes = raw; pl.kind = 1; return pl; }
  pl.tc = reduceType(zMin, dtZ, pl.dtRed);
  int nb = 1 + dtSize(pl.dtRed);
  const u32 maxElem = (e > 0) ? (u32)(mv + 0.5) : 0u;
  pl.maxElem = maxElem;
  bool lut = tryLut;
  if (maxElem > 0)
    nb += !tryLut ? (int)sizeSimple((u32)n, maxElem) : (int)sizeLut((u32)n, qMax, nDistinct - 1, lut);
  if (nb < raw) pl.kind = (maxElem == 0) ? 2 : (!lut ? 3 : 4);
  else { nb = raw; pl.kind = 1; }
  pl.nBytes = nb;
  return pl;
}

template<class Z> __device__ __forceinline__ u32 quantLossless(Z v, Z zMin) { return (u32)((i64)v - (i64)zMin); }
template<> __device__ __forceinline__ u32 quantLossless<u32>(u32 v, u32 zMin) { return v - zMin; }

}    // namespace lerc
