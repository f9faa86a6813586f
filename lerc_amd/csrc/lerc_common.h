// lerc_common.h -- definitions shared by host code and HIP kernels of the MI355X LERC path.
//
// Wire-format facts restated here follow the Esri/lerc reference (file:line relative to
// /root/reference): data types Lerc_types.h:22-32, header layout Lerc2.cpp:724-786, block flag /
// offset typing Lerc2.cpp:1949-2021 + Lerc2.h:457-542, BitStuffer2 stream BitStuffer2.cpp:35-153.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstring>

namespace lerc {

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;

enum ErrCode : u32 { kOk = 0, kFailed = 1, kWrongParam = 2, kBufferTooSmall = 3, kNaN = 4, kHasNoData = 5, kDimsTooLarge = 6 };
enum DataType : int { DT_Char = 0, DT_Byte, DT_Short, DT_UShort, DT_Int, DT_UInt, DT_Float, DT_Double, DT_Undefined };
enum ImageEncodeMode : int { IEM_Tiling = 0, IEM_DeltaHuffman = 1, IEM_Huffman = 2, IEM_DeltaDeltaHuffman = 3 };
enum BlockEncodeMode : int { BEM_Raw = 0, BEM_Simple = 1, BEM_Lut = 2 };

static const int kCodecVersion = 6;    // Lerc2.h:80

#define LERC_HD __host__ __device__ __forceinline__

template<class T> struct DtOf;
template<> struct DtOf<signed char>    { static const int v = DT_Char; };
template<> struct DtOf<unsigned char>  { static const int v = DT_Byte; };
template<> struct DtOf<short>          { static const int v = DT_Short; };
template<> struct DtOf<unsigned short> { static const int v = DT_UShort; };
template<> struct DtOf<int>            { static const int v = DT_Int; };
template<> struct DtOf<unsigned int>   { static const int v = DT_UInt; };
template<> struct DtOf<float>          { static const int v = DT_Float; };
template<> struct DtOf<double>         { static const int v = DT_Double; };

LERC_HD int dtSize(int dt)    // Lerc2.h:707-724
{
  return (dt <= DT_Byte) ? 1 : (dt <= DT_UShort) ? 2 : (dt <= DT_Float) ? 4 : (dt == DT_Double) ? 8 : 0;
}
LERC_HD u32 maxValToQuantize(int dt)    // Lerc2.h:685-703
{
  return (dt <= DT_UShort) ? ((1u << 15) - 1) : (dt <= DT_Double) ? ((1u << 30) - 1) : 0;
}
LERC_HD int bitLen(u32 v)    // number of bits of the largest element (BitStuffer2.cpp:41-43)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return 32 - __clz((int)v);
#else
  return v ? 32 - __builtin_clz(v) : 0;
#endif
}
LERC_HD int countFieldBytes(u32 n) { return n < 256 ? 1 : (n < 65536 ? 2 : 4); }    // BitStuffer2.h:64

// bytes of a BitStuffer2 "simple" stream: header byte + count field + ceil(n * numBits / 8)
// Codec 2 packs the n elements of nb bits MSB first into 32-bit words, shifts the last word down by the bytes it does
// not need and stores only the bytes in use (BitStuff_Before_Lerc2v3, BitStuffer2.cpp:292-351; same byte count as the later
// layout).  Seen as the little-endian bit stream of those bytes, element i is one or two runs of bits: its top n0 bits at
// bit pos0 and -- when it straddles two words -- its low n1 bits at pos1.
struct OldBitLayout { u32 pos0, n0, pos1, n1; };
LERC_HD OldBitLayout oldBitLayout(u32 i, int nb, u32 n)
{
  const u32 total = n * (u32)nb, lastWord = (total + 31u) / 32u - 1u;
  const u32 tailBits = total & 31u, tailBytes = (tailBits + 7u) >> 3;
  const u32 drop = tailBytes ? 8u * (4u - tailBytes) : 0u;    // NumTailBytesNotNeeded, BitStuffer2.h:127-132
  const u32 b = i * (u32)nb, w = b >> 5, s = b & 31u;
  OldBitLayout o;
  if (s + (u32)nb <= 32u)
  {
    o.n0 = (u32)nb; o.n1 = 0; o.pos1 = 0;
    o.pos0 = 32u * w + (32u - s - (u32)nb) - (w == lastWord ? drop : 0u);
  }
  else
  {
    o.n0 = 32u - s; o.pos0 = 32u * w;
    o.n1 = (u32)nb - o.n0;
    o.pos1 = 32u * (w + 1u) + (32u - o.n1) - (w + 1u == lastWord ? drop : 0u);
  }
  return o;
}

LERC_HD u32 sizeSimple(u32 n, u32 maxElem) { return 1 + countFieldBytes(n) + ((n * (u32)bitLen(maxElem) + 7) >> 3); }
// bytes of a LUT stream (BitStuffer2.cpp:262-287); nLut = number of distinct values minus one
LERC_HD u32 sizeLut(u32 n, u32 maxElem, u32 nLut, bool& doLut)
{
  int nb = bitLen(maxElem);
  u32 plain = 1 + countFieldBytes(n) + ((n * nb + 7) >> 3);
  int nbIdx = bitLen(nLut);
  u32 lut = 1 + countFieldBytes(n) + 1 + ((nLut * nb + 7) >> 3) + ((n * nbIdx + 7) >> 3);
  doLut = lut < plain;
  return lut < plain ? lut : plain;
}

// Offset ("zMin") type reduction, Lerc2.h:457-515.  Returns the 2-bit type code, sets dtRed.
template<class Z> LERC_HD int reduceType(Z z, int dt, int& dtRed)
{
  u8 b = (z >= 0 && z <= 255) ? (u8)z : 0;
  switch (dt)
  {
    case DT_Short:
    {
      signed char c = (z >= (double)-128 && z <= 127) ? (signed char)z : 0;
      int tc = (Z)c == z ? 2 : (Z)b == z ? 1 : 0;
      dtRed = dt - tc;
      return tc;
    }
    case DT_UShort:
    {
      int tc = (Z)b == z ? 1 : 0;
      dtRed = dt - 2 * tc;
      return tc;
    }
    case DT_Int:
    {
      short s = (z >= (double)SHRT_MIN && z <= SHRT_MAX) ? (short)z : 0;
      u16 us = (z >= 0 && z <= USHRT_MAX) ? (u16)z : 0;
      int tc = (Z)b == z ? 3 : (Z)s == z ? 2 : (Z)us == z ? 1 : 0;
      dtRed = dt - tc;
      return tc;
    }
    case DT_UInt:
    {
      u16 us = (z >= 0 && z <= USHRT_MAX) ? (u16)z : 0;
      int tc = (Z)b == z ? 2 : (Z)us == z ? 1 : 0;
      dtRed = dt - 2 * tc;
      return tc;
    }
    case DT_Float:
    {
      short s = (z >= (float)SHRT_MIN && z <= SHRT_MAX) ? (short)z : 0;
      int tc = (Z)b == z ? 2 : (Z)s == z ? 1 : 0;
      dtRed = tc == 0 ? dt : (tc == 1 ? DT_Short : DT_Byte);
      return tc;
    }
    case DT_Double:
    {
      short s = (z >= (double)SHRT_MIN && z <= SHRT_MAX) ? (short)z : 0;
      int l = (z >= (double)INT_MIN && z <= (double)INT_MAX) ? (int)z : 0;
      float f = (z >= -FLT_MAX && z <= FLT_MAX) ? (float)z : 0;
      int tc = (Z)s == z ? 3 : (Z)l == z ? 2 : (Z)f == z ? 1 : 0;
      dtRed = tc == 0 ? dt : dt - 2 * tc + 1;
      return tc;
    }
    default:
      dtRed = dt;
      return 0;
  }
}

LERC_HD int typeUsed(int dt, int tc)    // Lerc2.h:528-542
{
  int r;
  switch (dt)
  {
    case DT_Short: case DT_Int: r = dt - tc; break;
    case DT_UShort: case DT_UInt: r = dt - 2 * tc; break;
    case DT_Float: return tc == 0 ? dt : (tc == 1 ? DT_Short : DT_Byte);
    case DT_Double: r = tc == 0 ? dt : dt - 2 * tc + 1; break;
    default: return dt;
  }
  return (r >= DT_Char && r <= DT_Double) ? r : DT_Undefined;
}

// Little-endian typed store / load on (possibly unaligned) byte pointers; value goes through
// double exactly like Lerc2::WriteVariableDataType / ReadVariableDataType (Lerc2.h:546-681).
template<class P> LERC_HD void putBytes(P* dst, u64 bits, int n) { for (int i = 0; i < n; i++) dst[i] = (u8)(bits >> (8 * i)); }
template<class P> LERC_HD u64 getBytes(const P* src, int n) { u64 v = 0; for (int i = 0; i < n; i++) v |= (u64)src[i] << (8 * i); return v; }

LERC_HD u64 typedBits(double z, int dt)
{
  switch (dt)
  {
    case DT_Char:   return (u64)(u8)(signed char)z;
    case DT_Byte:   return (u64)(u8)z;
    case DT_Short:  return (u64)(u16)(short)z;
    case DT_UShort: return (u64)(u16)z;
    case DT_Int:    return (u64)(u32)(int)z;
    case DT_UInt:   return (u64)(u32)z;
    case DT_Float:  { float f = (float)z; u32 b; memcpy(&b, &f, 4); return b; }
    default:        { u64 b; memcpy(&b, &z, 8); return b; }
  }
}
LERC_HD double typedFromBits(u64 bits, int dt)
{
  switch (dt)
  {
    case DT_Char:   return (double)(signed char)(u8)bits;
    case DT_Byte:   return (double)(u8)bits;
    case DT_Short:  return (double)(short)(u16)bits;
    case DT_UShort: return (double)(u16)bits;
    case DT_Int:    return (double)(int)(u32)bits;
    case DT_UInt:   return (double)(u32)bits;
    case DT_Float:  { u32 b = (u32)bits; float f; memcpy(&f, &b, 4); return (double)f; }
    default:        { double d; memcpy(&d, &bits, 8); return d; }
  }
}

// validity bit of pixel k: MSB first within the byte (BitMask.h:67)
LERC_HD bool maskBit(const u8* bits, i64 k) { return (bits[k >> 3] & (0x80u >> (k & 7))) != 0; }

// ------------------------------------------------------------------------------------------------
// Parameters of one band handed to the block kernels (by value).
// ------------------------------------------------------------------------------------------------
struct BandParams
{
  int nRows, nCols, nDepth;
  int mb;              // micro block size (8 or 16 when encoding; <= 32 when decoding)
  int nTV, nTH;        // block rows / block columns
  int dt;              // DataType
  int version;         // codec version of the blob
  int allValid;        // numValidPixel == nRows * nCols  (no mask stored)
  int intLossless;     // dt < Float && maxZErr == 0.5        (Lerc2.cpp:1493-1494)
  int tryDiff;         // version >= 5 && nDepth > 1 && intLossless (Lerc2.cpp:1495)
  int checkOverflow;   // Lerc2.h:312-315
  u32 maxQ;            // maxValToQuantize(dt)
  double maxZErr;
  double scale;        // 1 / (2 * maxZErr)   (computed on the host; 0 when maxZErr == 0)
  double invScale;     // 2 * maxZErr
  double zMaxHdr;      // header zMax: decode clamp for nDepth == 1 (Lerc2.cpp:2112)
};

// Device-visible status word layout of a codec call.
struct DeviceStatus
{
  u32 error;           // first error code raised by a kernel (0 = none)
  u32 errorWhere;      // diagnostic: block / chunk index
  u32 pad[2];
};

// Result of the global statistics pre-pass (one per band; per-depth arrays live behind it).
struct BandStats
{
  u32 numValid;
  u32 hasNaN;          // any NaN at a valid pixel
  u32 notAllInt;       // some valid value is not an integer (float types)
  u32 mixedNaN;        // nDepth > 1: a pixel with some but not all values NaN
  u64 firstZeroIdx;    // reserved
  double raiseErr[9];  // TryRaiseMaxZError: max rounding error per candidate factor (Lerc2.cpp:1233-1318)
};

}    // namespace lerc
