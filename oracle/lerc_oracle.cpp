/*
 * oracle/lerc_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lerc_oracle.h).
 *
 * Sequential CPU restatement of the Lerc2 v6 codec path (reference: Esri/lerc @ LERC 4.2.0).
 * Every function cites the reference file:line whose behaviour it restates.  The code is organised
 * differently from the reference on purpose (byte cursors + free functions, one translation unit,
 * no class hierarchy) but must produce byte-identical blobs and bit-identical decodes; that is
 * enforced by tests/test_oracle_vs_reference.py against oracle/_ref/libLercRef.so.
 *
 * Build: make -C oracle      (g++ -O3 -ffp-contract=off; the reference's Release build has no
 *                             -march, i.e. no FMA contraction -- SURVEY.md App. B-2)
 */
#include "lerc_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <type_traits>
#include <utility>
#include <vector>

namespace orc {

typedef unsigned char u8;

enum Err { OK = 0, FAILED = 1, WRONG_PARAM = 2, BUFFER_TOO_SMALL = 3, ERR_NAN = 4, HAS_NODATA = 5, DIMS_TOO_LARGE = 6 };
// Lerc_types.h:22-32
enum DT { DT_CHAR = 0, DT_BYTE, DT_SHORT, DT_USHORT, DT_INT, DT_UINT, DT_FLOAT, DT_DOUBLE, DT_UNDEF };
// Lerc2.h:141-142
enum ImageMode { IEM_TILING = 0, IEM_DELTA_HUFFMAN = 1, IEM_HUFFMAN = 2, IEM_DELTADELTA_HUFFMAN = 3 };
enum BlockMode { BEM_RAW = 0, BEM_SIMPLE = 1, BEM_LUT = 2 };

static const int kCurrentVersion = 6;    // Lerc2.h:80

template<class T> struct DtOf;
template<> struct DtOf<signed char>    { static const DT v = DT_CHAR; };
template<> struct DtOf<unsigned char>  { static const DT v = DT_BYTE; };
template<> struct DtOf<short>          { static const DT v = DT_SHORT; };
template<> struct DtOf<unsigned short> { static const DT v = DT_USHORT; };
template<> struct DtOf<int>            { static const DT v = DT_INT; };
template<> struct DtOf<unsigned int>   { static const DT v = DT_UINT; };
template<> struct DtOf<float>          { static const DT v = DT_FLOAT; };
template<> struct DtOf<double>         { static const DT v = DT_DOUBLE; };

// Lerc2.h:707-724
static unsigned dtSize(int dt)
{
  switch (dt) {
    case DT_CHAR: case DT_BYTE: return 1;
    case DT_SHORT: case DT_USHORT: return 2;
    case DT_INT: case DT_UINT: case DT_FLOAT: return 4;
    case DT_DOUBLE: return 8;
    default: return 0;
  }
}

// Lerc2.h:685-703
static unsigned maxValToQuantize(int dt)
{
  switch (dt) {
    case DT_CHAR: case DT_BYTE: case DT_SHORT: case DT_USHORT: return (1u << 15) - 1;
    case DT_INT: case DT_UINT: case DT_FLOAT: case DT_DOUBLE: return (1u << 30) - 1;
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------------------
// byte cursors
// ---------------------------------------------------------------------------------------------
struct Reader
{
  const u8* p;
  size_t left;
  bool get(void* dst, size_t n)
  {
    if (left < n) return false;
    memcpy(dst, p, n);
    p += n; left -= n;
    return true;
  }
  bool skip(size_t n) { if (left < n) return false; p += n; left -= n; return true; }
};

struct Writer
{
  u8* p;
  void put(const void* src, size_t n) { memcpy(p, src, n); p += n; }
  void byte(u8 b) { *p++ = b; }
};

// ---------------------------------------------------------------------------------------------
// Fletcher32 variant -- Lerc2.cpp:1037-1064.  Words are big-endian byte pairs, sums start 0xffff,
// folded every 359 words; an odd trailing byte counts as (b << 8).
// ---------------------------------------------------------------------------------------------
static unsigned fletcher32(const u8* b, int len)
{
  unsigned s1 = 0xffff, s2 = 0xffff;
  int words = len / 2;
  int done = 0;
  while (done < words)
  {
    int n = std::min(359, words - done);
    for (int i = 0; i < n; i++, b += 2)
    {
      s1 += ((unsigned)b[0] << 8);
      s1 += b[1];
      s2 += s1;
    }
    done += n;
    s1 = (s1 & 0xffff) + (s1 >> 16);
    s2 = (s2 & 0xffff) + (s2 >> 16);
  }
  if (len & 1) { s1 += ((unsigned)b[0] << 8); s2 += s1; }
  s1 = (s1 & 0xffff) + (s1 >> 16);
  s2 = (s2 & 0xffff) + (s2 >> 16);
  return (s2 << 16) | s1;
}

// ---------------------------------------------------------------------------------------------
// validity bit mask -- BitMask.h:67: pixel k <-> bits[k >> 3] & (0x80 >> (k & 7))
// ---------------------------------------------------------------------------------------------
struct Mask
{
  int nCols = 0, nRows = 0;
  std::vector<u8> bits;
  size_t nBytes() const { return ((size_t)nCols * nRows + 7) >> 3; }
  void resize(int c, int r) { if (c != nCols || r != nRows) { nCols = c; nRows = r; bits.assign(nBytes(), 0); } }
  void fill(bool v) { std::fill(bits.begin(), bits.end(), v ? 255 : 0); }
  bool valid(int64_t k) const { return (bits[k >> 3] & (0x80 >> (k & 7))) != 0; }
  void clear(int64_t k) { bits[k >> 3] &= (u8)~(0x80 >> (k & 7)); }
  // BitMask.cpp:93-112 -- popcount minus the undefined tail bits of the last byte
  int64_t countValid() const
  {
    int64_t s = 0;
    for (u8 b : bits) s += __builtin_popcount(b);
    int64_t total = (int64_t)nBytes() * 8;
    for (int64_t k = (int64_t)nCols * nRows; k < total; k++) if (valid(k)) s--;
    return s;
  }
};

// ---------------------------------------------------------------------------------------------
// RLE for the mask bytes -- RLE.cpp:32-119 (size), :123-243 (compress), :296-331 (decompress).
// Stream = [int16 count][payload]...; count > 0: that many literal bytes; count < 0: one byte
// repeated -count times; -32768 terminates.  A run needs >= 5 equal bytes (RLE.h:45).
// The encoder below is a direct state machine over (literal, run) modes; it emits segments as it
// goes and so also serves as the size calculator.
// ---------------------------------------------------------------------------------------------
static const int kMinRun = 5;

static void rleEncode(const u8* src, size_t n, std::vector<u8>& out)
{
  out.clear();
  auto putCount = [&](size_t at, int v) { short s = (short)v; memcpy(&out[at], &s, 2); };
  size_t cntPos = 0;             // where the pending segment's count goes
  out.resize(2);
  size_t lit = 0, run = 0;
  bool litMode = true;
  auto closeSeg = [&](int count) { putCount(cntPos, count); cntPos = out.size(); out.resize(out.size() + 2); };

  size_t i = 0;
  for (; i + 1 < n; i++)
  {
    u8 c = src[i];
    if (c != src[i + 1])
    {
      out.push_back(c);
      if (litMode) lit++;
      else { run++; closeSeg(-(int)run); litMode = true; lit = 0; run = 0; }
    }
    else if (!litMode) run++;
    else
    {
      bool enough = false;
      if (i + kMinRun < n)
      {
        int k = 1;
        while (k < kMinRun && src[i + k] == c) k++;
        enough = (k >= kMinRun);
      }
      if (!enough) { out.push_back(c); lit++; }
      else
      {
        if (lit > 0) closeSeg((int)lit);
        litMode = false; lit = 0; run = 1;
      }
    }
    if (lit == 32767) { closeSeg(32767); lit = 0; }
    if (run == 32767) { out.push_back(c); closeSeg(-32767); run = 0; }
  }
  out.push_back(src[i]);    // last byte
  if (litMode) closeSeg((int)(lit + 1));
  else closeSeg(-(int)(run + 1));
  putCount(cntPos, -32768);
}

static bool rleDecode(const u8* src, size_t left, u8* dst, size_t dstSize)
{
  if (!src || !dst || left < 2) return false;
  size_t at = 0;
  left -= 2;
  short cnt; memcpy(&cnt, src, 2); src += 2;
  while (cnt != -32768)
  {
    int n = cnt <= 0 ? -cnt : cnt;
    size_t m = cnt <= 0 ? 1 : (size_t)n;
    if (left < m + 2 || at + n > dstSize) return false;
    if (cnt > 0) { memcpy(dst + at, src, n); src += n; }
    else { memset(dst + at, *src, n); src += 1; }
    at += n;
    left -= m + 2;
    memcpy(&cnt, src, 2); src += 2;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Header -- Lerc2.h:102-131, write Lerc2.cpp:724-786, read :790-917, size :710-720
// ---------------------------------------------------------------------------------------------
struct Header
{
  int version = kCurrentVersion;
  unsigned checksum = 0;
  int nRows = 0, nCols = 0, nDepth = 1, numValid = 0, mbSize = 8, blobSize = 0, dt = DT_UNDEF, nBlobsMore = 0;
  u8 passNoData = 0, isInt = 0, rsv3 = 0, rsv4 = 0;
  double maxZErr = 0, zMin = 0, zMax = 0, noDataVal = 0, noDataValOrig = 0;

  bool tryHuffmanInt() const { return version >= 2 && (dt == DT_BYTE || dt == DT_CHAR) && maxZErr == 0.5; }   // Lerc2.h:129
  bool tryHuffmanFlt() const { return version >= 6 && (dt == DT_FLOAT || dt == DT_DOUBLE) && maxZErr == 0; }  // Lerc2.h:130
};

static unsigned headerBytes(int v)
{
  return 6 + 4 + (v >= 3 ? 4 : 0) + 4 * (v >= 4 ? 7 : 6) + (v >= 6 ? 4 + 4 : 0) + 8 * (v >= 6 ? 5 : 3);
}

static void writeHeader(Writer& w, const Header& h)
{
  w.put("Lerc2 ", 6);
  w.put(&h.version, 4);
  if (h.version >= 3) { unsigned zero = 0; w.put(&zero, 4); }
  w.put(&h.nRows, 4);
  w.put(&h.nCols, 4);
  if (h.version >= 4) w.put(&h.nDepth, 4);
  w.put(&h.numValid, 4);
  w.put(&h.mbSize, 4);
  w.put(&h.blobSize, 4);
  w.put(&h.dt, 4);
  if (h.version >= 6)
  {
    w.put(&h.nBlobsMore, 4);
    w.byte(h.passNoData); w.byte(h.isInt); w.byte(h.rsv3); w.byte(h.rsv4);
  }
  w.put(&h.maxZErr, 8);
  w.put(&h.zMin, 8);
  w.put(&h.zMax, 8);
  if (h.version >= 6) { w.put(&h.noDataVal, 8); w.put(&h.noDataValOrig, 8); }
}

static bool readHeader(Reader& r0, Header& h)
{
  Reader r = r0;
  h = Header();
  char key[6];
  if (!r.get(key, 6) || memcmp(key, "Lerc2 ", 6)) return false;
  if (!r.get(&h.version, 4)) return false;
  if (h.version < 0 || h.version > kCurrentVersion) return false;
  if (h.version >= 3 && !r.get(&h.checksum, 4)) return false;
  h.nDepth = 1;
  if (!r.get(&h.nRows, 4) || !r.get(&h.nCols, 4)) return false;
  if (h.version >= 4 && !r.get(&h.nDepth, 4)) return false;
  if (!r.get(&h.numValid, 4) || !r.get(&h.mbSize, 4) || !r.get(&h.blobSize, 4) || !r.get(&h.dt, 4)) return false;
  if (h.version >= 6)
  {
    u8 b[4];
    if (!r.get(&h.nBlobsMore, 4) || !r.get(b, 4)) return false;
    h.passNoData = b[0]; h.isInt = b[1]; h.rsv3 = b[2]; h.rsv4 = b[3];
  }
  if (!r.get(&h.maxZErr, 8) || !r.get(&h.zMin, 8) || !r.get(&h.zMax, 8)) return false;
  if (h.version >= 6 && (!r.get(&h.noDataVal, 8) || !r.get(&h.noDataValOrig, 8))) return false;

  if (h.nRows <= 0 || h.nCols <= 0 || h.nDepth <= 0 || h.numValid < 0 || h.mbSize <= 0 || h.blobSize <= 0
    || h.dt < DT_CHAR || h.dt > DT_DOUBLE)
    return false;

  const uint64_t nPix = (uint64_t)h.nRows * h.nCols, lim = (uint64_t)INT_MAX, bpp = dtSize(h.dt);
  if (nPix > lim || (uint64_t)h.numValid > nPix) return false;
  if (h.mbSize > 32 || bpp * h.nDepth > lim || bpp * h.nDepth * nPix > lim) return false;
  r0 = r;
  return true;
}

// Lerc2.cpp:495-512
static bool peekHeader(const u8* p, size_t n, Header& h, bool& hasMask)
{
  if (!p) return false;
  Reader r{ p, n };
  if (!readHeader(r, h)) return false;
  int nm = 0;
  if (r.left < 4) return false;
  memcpy(&nm, r.p, 4);
  if (nm < 0) return false;
  hasMask = nm > 0;
  return true;
}

// ---------------------------------------------------------------------------------------------
// BitStuffer2 (v3+ layout) -- BitStuffer2.cpp:35-75 (simple), :79-153 (LUT), :159-258 (decode),
// :432-472 (stuff), :476-540 (unstuff); size formulas BitStuffer2.h:68-74, BitStuffer2.cpp:262-287.
// Element i occupies bits [i*nb, (i+1)*nb) of a little-endian bit stream, byte length ceil(n*nb/8).
// ---------------------------------------------------------------------------------------------
static int bitLen(unsigned v) { int n = 0; while (n < 32 && (v >> n)) n++; return n; }
static int countBytes(unsigned k) { return k < 256 ? 1 : (k < 65536 ? 2 : 4); }

static void stuffBits(Writer& w, const unsigned* v, unsigned n, int nb)
{
  size_t nBytes = ((size_t)n * nb + 7) >> 3;
  memset(w.p, 0, nBytes);
  uint64_t bit = 0;
  for (unsigned i = 0; i < n; i++, bit += nb)
  {
    uint64_t x = (uint64_t)v[i] << (bit & 7);
    size_t at = bit >> 3;
    for (int k = 0; x; k++, x >>= 8) w.p[at + k] |= (u8)x;
  }
  w.p += nBytes;
}

// Codec 2 (BitStuff_Before_Lerc2v3 / BitUnStuff_Before_Lerc2v3, BitStuffer2.cpp:292-425): the elements are packed MSB
// first into 32-bit words, the last word is shifted down by the bytes it does not need and only its used bytes are kept.
static unsigned tailBytesNotNeeded(unsigned n, int nb)    // BitStuffer2.h:127-132
{
  int numBitsTail = (int)(((unsigned long long)n * nb) & 31);
  int numBytesTail = (numBitsTail + 7) >> 3;
  return (numBytesTail > 0) ? 4 - numBytesTail : 0;
}

static void stuffBitsOld(Writer& w, const unsigned* v, unsigned n, int nb)
{
  const unsigned numUInts = (unsigned)(((unsigned long long)n * nb + 31) / 32);
  std::vector<unsigned> arr(numUInts, 0u);
  unsigned* dst = arr.data();
  int bitPos = 0;
  for (unsigned i = 0; i < n; i++)
  {
    if (32 - bitPos >= nb)
    {
      *dst |= v[i] << (32 - bitPos - nb);
      bitPos += nb;
      if (bitPos == 32) { bitPos = 0; dst++; }
    }
    else
    {
      const int k = nb - (32 - bitPos);
      *dst++ |= v[i] >> k;
      *dst |= v[i] << (32 - k);
      bitPos = k;
    }
  }
  const unsigned drop = tailBytesNotNeeded(n, nb);
  for (unsigned k = drop; k; --k) arr[numUInts - 1] >>= 8;
  const size_t nBytes = (size_t)numUInts * 4 - drop;
  memcpy(w.p, arr.data(), nBytes);
  w.p += nBytes;
}

static bool unstuffBitsOld(Reader& r, std::vector<unsigned>& v, unsigned n, int nb)
{
  if (n == 0 || nb >= 32) return false;
  const size_t numUInts = (size_t)(((unsigned long long)n * nb + 31) / 32);
  const unsigned drop = tailBytesNotNeeded(n, nb);
  const size_t nBytes = ((size_t)n * nb + 7) / 8;
  if (r.left + drop < numUInts * 4 || r.left < nBytes) return false;
  std::vector<unsigned> arr(numUInts, 0u);
  memcpy(arr.data(), r.p, nBytes);
  for (unsigned k = drop; k; --k) arr[numUInts - 1] <<= 8;
  v.assign(n, 0u);
  const unsigned* src = arr.data();
  int bitPos = 0;
  for (unsigned i = 0; i < n; i++)
  {
    if (32 - bitPos >= nb)
    {
      v[i] = (*src << bitPos) >> (32 - nb);
      bitPos += nb;
      if (bitPos == 32) { bitPos = 0; src++; }
    }
    else
    {
      v[i] = (*src++ << bitPos) >> (32 - nb);
      bitPos -= (32 - nb);
      v[i] |= *src >> (32 - bitPos);
    }
  }
  r.skip(nBytes);
  return true;
}

static bool unstuffBits(Reader& r, std::vector<unsigned>& v, unsigned n, int nb, int lercVersion = kCurrentVersion)
{
  if (lercVersion < 3) return unstuffBitsOld(r, v, n, nb);
  if (n == 0 || nb >= 32) return false;
  size_t nBytes = ((size_t)n * nb + 7) >> 3;
  if (r.left < nBytes) return false;
  v.resize(n);
  uint64_t bit = 0;
  const unsigned mask = (1u << nb) - 1;    // 0 < nb < 32
  for (unsigned i = 0; i < n; i++, bit += nb)
  {
    size_t at = bit >> 3;
    uint64_t x = 0;
    for (size_t k = 0; k < 5 && at + k < nBytes; k++) x |= (uint64_t)r.p[at + k] << (8 * k);
    v[i] = (unsigned)(x >> (bit & 7)) & mask;
  }
  r.skip(nBytes);
  return true;
}

static unsigned sizeSimple(unsigned n, unsigned maxElem)
{
  return 1 + countBytes(n) + ((n * bitLen(maxElem) + 7) >> 3);
}

static void putCountField(Writer& w, unsigned k, int nBytes)
{
  if (nBytes == 1) w.byte((u8)k);
  else if (nBytes == 2) { unsigned short s = (unsigned short)k; w.put(&s, 2); }
  else w.put(&k, 4);
}

static bool encodeSimple(Writer& w, const std::vector<unsigned>& v, int lercVersion = kCurrentVersion)
{
  if (v.empty()) return false;
  unsigned mx = *std::max_element(v.begin(), v.end());
  int nb = bitLen(mx);
  if (nb >= 32) return false;
  unsigned n = (unsigned)v.size();
  int cb = countBytes(n);
  int code = (cb == 4) ? 0 : 3 - cb;
  w.byte((u8)(nb | (code << 6)));
  putCountField(w, n, cb);
  if (nb > 0) { if (lercVersion >= 3) stuffBits(w, v.data(), n, nb); else stuffBitsOld(w, v.data(), n, nb); }
  return true;
}

typedef std::pair<unsigned, unsigned> QIdx;    // (quantized value, element index)

// BitStuffer2.cpp:262-287
static unsigned sizeLut(const std::vector<QIdx>& sorted, bool& doLut)
{
  unsigned mx = sorted.back().first, n = (unsigned)sorted.size();
  int nb = bitLen(mx);
  unsigned plain = 1 + countBytes(n) + ((n * nb + 7) >> 3);
  int nLut = 0;
  for (unsigned i = 1; i < n; i++) if (sorted[i].first != sorted[i - 1].first) nLut++;
  int nbIdx = 0;
  while (nLut >> nbIdx) nbIdx++;
  unsigned lut = 1 + countBytes(n) + 1 + (((unsigned)nLut * nb + 7) >> 3) + ((n * nbIdx + 7) >> 3);
  doLut = lut < plain;
  return std::min(lut, plain);
}

// BitStuffer2.cpp:79-153
static bool encodeLut(Writer& w, const std::vector<QIdx>& sorted, int lercVersion = kCurrentVersion)
{
  if (sorted.empty() || sorted[0].first != 0) return false;
  unsigned n = (unsigned)sorted.size();
  std::vector<unsigned> lut, idx(n, 0);
  unsigned cur = 0;
  for (unsigned i = 1; i < n; i++)
  {
    idx[sorted[i - 1].second] = cur;
    if (sorted[i].first != sorted[i - 1].first) { lut.push_back(sorted[i].first); cur++; }
  }
  idx[sorted[n - 1].second] = cur;
  if (lut.empty()) return false;
  int nb = bitLen(lut.back());
  if (nb <= 0 || nb >= 32) return false;
  int cb = countBytes(n);
  int code = (cb == 4) ? 0 : 3 - cb;
  w.byte((u8)(nb | (code << 6) | (1 << 5)));
  putCountField(w, n, cb);
  unsigned nLut = (unsigned)lut.size();
  if (nLut < 1 || nLut >= 255) return false;
  w.byte((u8)(nLut + 1));
  if (lercVersion >= 3) stuffBits(w, lut.data(), nLut, nb); else stuffBitsOld(w, lut.data(), nLut, nb);
  int nbIdx = 0;
  while (nLut >> nbIdx) nbIdx++;
  if (lercVersion >= 3) stuffBits(w, idx.data(), n, nbIdx); else stuffBitsOld(w, idx.data(), n, nbIdx);
  return true;
}

// BitStuffer2.cpp:159-258 (v3+ branch only).  `out` keeps its previous contents when numBits == 0,
// exactly like the reference's reused buffer.
static bool decodeBitStuffer(Reader& r, std::vector<unsigned>& out, size_t maxCount, int lercVersion)
{
  u8 b0;
  if (!r.get(&b0, 1)) return false;
  int code = b0 >> 6;
  int cb = (code == 0) ? 4 : 3 - code;
  bool lutMode = (b0 & 32) != 0;
  int nb = b0 & 31;
  unsigned n = 0;
  if (cb == 1) { u8 c; if (!r.get(&c, 1)) return false; n = c; }
  else if (cb == 2) { unsigned short s; if (!r.get(&s, 2)) return false; n = s; }
  else if (cb == 4) { if (!r.get(&n, 4)) return false; }
  else return false;
  if (n > maxCount) return false;

  if (!lutMode)
  {
    if (nb > 0 && !unstuffBits(r, out, n, nb, lercVersion)) return false;
    return true;
  }
  if (nb == 0) return false;
  u8 lb;
  if (!r.get(&lb, 1)) return false;
  int nLut = lb - 1;
  std::vector<unsigned> lut;
  if (nLut < 1) return false;
  if (!unstuffBits(r, lut, (unsigned)nLut, nb, lercVersion)) return false;
  int nbIdx = 0;
  while (nLut >> nbIdx) nbIdx++;
  if (nbIdx == 0) return false;
  if (!unstuffBits(r, out, n, nbIdx, lercVersion)) return false;
  lut.insert(lut.begin(), 0u);
  for (unsigned i = 0; i < n; i++)
  {
    if (out[i] >= lut.size()) return false;    // reference relies on the checksum here (:235-237)
    out[i] = lut[out[i]];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Huffman -- Huffman.cpp:35-81 (tree), :541-572 (canonical), :383-438 (range), :357-379 + :85-111
// (sizes), :126-166 / :442-467 (table write), :170-234 / :471-537 (table read), Huffman.h:218-255
// (MSB-first push into little-endian uint32 words).
// ---------------------------------------------------------------------------------------------
typedef std::pair<unsigned short, unsigned> HCode;    // (length, code)

struct HeapItem
{
  int weight;    // = -count (Huffman.h:72)
  int node;
  bool operator<(const HeapItem& o) const { return weight < o.weight; }
};

// Tie-breaking must match the reference, which pops from a std::priority_queue keyed on weight
// only (Huffman.cpp:40-61).  Using the same container with the same push/pop sequence reproduces
// its heap order exactly.
static bool huffBuild(const std::vector<int>& histo, std::vector<HCode>& table)
{
  int size = (int)histo.size();
  if (size == 0 || size >= (1 << 15)) return false;
  struct N { int c0, c1, sym; };
  std::vector<N> nodes;
  std::priority_queue<HeapItem> pq;
  for (int i = 0; i < size; i++)
    if (histo[i] > 0) { nodes.push_back({ -1, -1, i }); pq.push({ -histo[i], (int)nodes.size() - 1 }); }
  if (pq.size() < 2) return false;
  while (pq.size() > 1)
  {
    HeapItem a = pq.top(); pq.pop();
    HeapItem b = pq.top(); pq.pop();
    nodes.push_back({ a.node, b.node, -1 });
    pq.push({ a.weight + b.weight, (int)nodes.size() - 1 });
  }
  table.assign(size, HCode(0, 0));
  // depth-first length assignment; > 32 bits is refused (Huffman.h:84-99)
  std::vector<std::pair<int, int> > st;
  st.push_back({ pq.top().node, 0 });
  while (!st.empty())
  {
    std::pair<int, int> t = st.back(); st.pop_back();
    const N& nd = nodes[t.first];
    if (nd.c0 >= 0)
    {
      if (t.second == 32) return false;
      st.push_back({ nd.c0, t.second + 1 });
      st.push_back({ nd.c1, t.second + 1 });
    }
    else table[nd.sym].first = (unsigned short)t.second;
  }
  // canonical codes: sort by (len * size - index) descending, assign increasing codes, shifting
  // right when the length drops (Huffman.cpp:541-572)
  std::vector<std::pair<int, unsigned> > order(size, std::pair<int, unsigned>(0, 0));
  for (int i = 0; i < size; i++)
    if (table[i].first > 0) order[i] = std::pair<int, unsigned>(table[i].first * size - i, (unsigned)i);
  std::sort(order.begin(), order.end(),
    [](const std::pair<int, unsigned>& a, const std::pair<int, unsigned>& b) { return a.first > b.first; });
  unsigned short len = table[order[0].second].first;
  unsigned code = 0;
  for (int i = 0; i < size && order[i].first > 0; i++)
  {
    unsigned k = order[i].second;
    int delta = len - table[k].first;
    code >>= delta;
    len = (unsigned short)(len - delta);
    table[k].second = code++;
  }
  return true;
}

static int wrapIdx(int i, int size) { return i - (i < size ? 0 : size); }

static bool huffRange(const std::vector<HCode>& t, int& i0, int& i1, int& maxLen)
{
  int size = (int)t.size();
  if (size == 0 || size >= (1 << 15)) return false;
  int i = 0;
  while (i < size && t[i].first == 0) i++;
  i0 = i;
  i = size - 1;
  while (i >= 0 && t[i].first == 0) i--;
  i1 = i + 1;
  if (i1 <= i0) return false;
  int bestStart = 0, bestLen = 0;
  for (int j = 0; j < size;)
  {
    while (j < size && t[j].first > 0) j++;
    int k0 = j;
    while (j < size && t[j].first == 0) j++;
    if (j - k0 > bestLen) { bestStart = k0; bestLen = j - k0; }
  }
  if (size - bestLen < i1 - i0) { i0 = bestStart + bestLen; i1 = bestStart + size; }
  if (i1 <= i0) return false;
  int m = 0;
  for (int k = i0; k < i1; k++) m = std::max(m, (int)t[wrapIdx(k, size)].first);
  if (m <= 0 || m > 32) return false;
  maxLen = m;
  return true;
}

// MSB-first bit sink over little-endian uint32 words (Huffman.h:218-255).  Words are zeroed when
// first touched; the caller advances past the partial word itself.
struct HuffSink
{
  u8* p;
  int bitPos;
  void push(unsigned v, int len)
  {
    unsigned w;
    if (32 - bitPos >= len)
    {
      if (bitPos == 0) memset(p, 0, 4);
      memcpy(&w, p, 4);
      w |= v << (32 - bitPos - len);
      memcpy(p, &w, 4);
      bitPos += len;
      if (bitPos == 32) { bitPos = 0; p += 4; }
    }
    else
    {
      bitPos += len - 32;
      memcpy(&w, p, 4);
      w |= v >> bitPos;
      memcpy(p, &w, 4);
      p += 4;
      w = v << (32 - bitPos);
      memcpy(p, &w, 4);
    }
  }
};

static bool huffTableBytes(const std::vector<HCode>& t, int& nBytes)
{
  int i0, i1, maxLen;
  if (!huffRange(t, i0, i1, maxLen)) return false;
  int size = (int)t.size(), sum = 0;
  for (int i = i0; i < i1; i++) sum += t[wrapIdx(i, size)].first;
  nBytes = 16 + (int)sizeSimple((unsigned)(i1 - i0), (unsigned)maxLen) + 4 * ((((sum + 7) >> 3) + 3) >> 2);
  return true;
}

// Huffman.cpp:85-111
static bool huffCompressedBytes(const std::vector<HCode>& t, const std::vector<int>& histo, int& nBytes)
{
  if (!huffTableBytes(t, nBytes)) return false;
  int bits = 0, elems = 0;
  for (size_t i = 0; i < histo.size(); i++)
    if (histo[i] > 0) { bits += histo[i] * t[i].first; elems += histo[i]; }
  if (elems == 0) return false;
  nBytes += 4 * (((((bits + 7) >> 3) + 3) >> 2) + 1);
  return true;
}

static bool huffWriteTable(Writer& w, const std::vector<HCode>& t, int lercVersion = kCurrentVersion)
{
  int i0, i1, maxLen;
  if (!huffRange(t, i0, i1, maxLen)) return false;
  int size = (int)t.size();
  std::vector<unsigned> lens(i1 - i0);
  for (int i = i0; i < i1; i++) lens[i - i0] = t[wrapIdx(i, size)].first;
  int hdr[4] = { 4, size, i0, i1 };
  w.put(hdr, 16);
  if (!encodeSimple(w, lens, lercVersion)) return false;
  HuffSink s{ w.p, 0 };
  for (int i = i0; i < i1; i++)
  {
    const HCode& c = t[wrapIdx(i, size)];
    if (c.first > 0) s.push(c.second, c.first);
  }
  w.p = s.p + (s.bitPos > 0 ? 4 : 0);
  return true;
}

static bool huffReadTable(Reader& r0, std::vector<HCode>& t, int lercVersion)
{
  Reader r = r0;
  int hdr[4];
  if (!r.get(hdr, 16)) return false;
  if (hdr[0] < 2) return false;
  const int size = hdr[1], i0 = hdr[2], i1 = hdr[3];
  if (i0 >= i1 || i0 < 0 || size < 0 || size > (1 << 15)) return false;
  if (wrapIdx(i0, size) >= size || wrapIdx(i1 - 1, size) >= size) return false;
  std::vector<unsigned> lens(i1 - i0, 0);
  if (!decodeBitStuffer(r, lens, lens.size(), lercVersion)) return false;
  if (lens.size() != (size_t)(i1 - i0)) return false;
  t.assign(size, HCode(0, 0));
  for (int i = i0; i < i1; i++) t[wrapIdx(i, size)].first = (unsigned short)lens[i - i0];

  // codes, MSB-first (Huffman.cpp:471-537)
  const u8* p0 = r.p;
  const u8* p = p0;
  size_t left = r.left;
  int bitPos = 0;
  for (int i = i0; i < i1; i++)
  {
    int k = wrapIdx(i, size);
    int len = t[k].first;
    if (len == 0) continue;
    if (left < 4 || len > 32) return false;
    unsigned w;
    memcpy(&w, p, 4);
    t[k].second = (w << bitPos) >> (32 - len);
    if (32 - bitPos >= len)
    {
      bitPos += len;
      if (bitPos == 32) { bitPos = 0; p += 4; left -= 4; }
    }
    else
    {
      bitPos += len - 32;
      p += 4; left -= 4;
      if (left < 4) return false;
      memcpy(&w, p, 4);
      t[k].second |= w >> (32 - bitPos);
    }
  }
  size_t used = (size_t)(p - p0) + (bitPos > 0 ? 4 : 0);
  if (r.left < used) return false;
  r.skip(used);
  r0 = r;
  return true;
}

// Prefix decoder for the pixel stream.  The reference uses a 12-bit LUT plus a tree for longer
// codes (Huffman.cpp:238-330, Huffman.h:144-214); for a prefix-free table any correct decoder
// returns the same symbols, so this restatement matches codes length by length.  The stream
// position bookkeeping (whole uint32 words, >= 4 bytes must remain) follows DecodeOneValue.
struct HuffDecoder
{
  std::vector<std::vector<std::pair<unsigned, int> > > byLen;    // [len] -> sorted (code, symbol)
  int maxLen = 0;
  bool init(const std::vector<HCode>& t)
  {
    int i0, i1;
    if (!huffRange(t, i0, i1, maxLen)) return false;
    byLen.assign(33, {});
    for (size_t k = 0; k < t.size(); k++)
      if (t[k].first > 0) byLen[t[k].first].push_back({ t[k].second, (int)k });
    for (auto& v : byLen) std::sort(v.begin(), v.end());
    return true;
  }
  bool next(const u8*& p, size_t& left, int& bitPos, int& sym) const
  {
    if (left < 4) return false;
    // gather up to 64 bits MSB-first starting at bitPos
    unsigned w0, w1 = 0;
    memcpy(&w0, p, 4);
    if (left >= 8) memcpy(&w1, p + 4, 4);
    uint64_t window = ((uint64_t)w0 << 32) | w1;
    window <<= bitPos;
    for (int len = 1; len <= maxLen; len++)
    {
      const auto& v = byLen[len];
      if (v.empty()) continue;
      unsigned code = (unsigned)(window >> (64 - len));
      auto it = std::lower_bound(v.begin(), v.end(), std::pair<unsigned, int>(code, INT_MIN));
      if (it != v.end() && it->first == code)
      {
        if (bitPos + len > 32 && left < 8) return false;
        sym = it->second;
        bitPos += len;
        if (bitPos >= 32) { bitPos -= 32; p += 4; left -= 4; }
        return true;
      }
    }
    return false;
  }
};

// ---------------------------------------------------------------------------------------------
// Lossless float / double (image mode IEM_DeltaDeltaHuffman) -- fpl_Lerc2Ext.cpp, fpl_UnitTypes.cpp,
// fpl_Predictor.cpp, fpl_Compression.cpp, fpl_EsriHuffman.cpp.  Float bits are reordered to
// exponent|sign|mantissa, a predictor (none / row differences / row + column differences, mantissa and the bits
// above it differenced apart) is picked from entropy estimates over sample blocks, the result is split into byte
// planes, every plane gets a byte-wise difference order picked the same way, and is then stored as one value,
// PackBits, raw bytes or Huffman codes -- whichever is smallest.
// ---------------------------------------------------------------------------------------------
static const int kFplPrime = 7;      // fpl_Compression.h:33
static const int kFplMaxDelta = 5;   // fpl_Predictor.h:33

static uint32_t fplFwd32(uint32_t a) { return (a & 0x007FFFFFu) | (((a >> 23) & 0xFFu) << 24) | ((a >> 31) << 23); }              // fpl_UnitTypes.cpp:39-51
static uint32_t fplBack32(uint32_t a) { return (a & 0x007FFFFFu) | (((a >> 24) & 0xFFu) << 23) | (((a >> 23) & 1u) << 31); }      // :53-65
static uint32_t fplSub(uint32_t a, uint32_t b) { return ((a - b) & 0x007FFFFFu) | ((((a >> 23) - (b >> 23)) & 0x1FFu) << 23); }  // :83-97
static uint32_t fplAdd(uint32_t a, uint32_t b) { return ((a + b) & 0x007FFFFFu) | ((((a >> 23) + (b >> 23)) & 0x1FFu) << 23); }  // :99-113
static uint64_t fplSub(uint64_t a, uint64_t b)    // :119-136
{
  return ((a - b) & 0x000FFFFFFFFFFFFFull) | ((((a >> 52) - (b >> 52)) & 0xFFFull) << 52);
}
static uint64_t fplAdd(uint64_t a, uint64_t b)    // :138-155
{
  return ((a + b) & 0x000FFFFFFFFFFFFFull) | ((((a >> 52) + (b >> 52)) & 0xFFFull) << 52);
}

// fpl_Compression.cpp:85-113
static long fplEntropy(const u8* p, size_t size)
{
  unsigned long table[256];
  memset(table, 0, sizeof(table));
  int total = 0;
  for (size_t i = 0; i < size; i += kFplPrime) { table[p[i]]++; total++; }
  double totalBits = 0;
  for (int i = 0; i < 256; i++)
  {
    if (table[i] == 0) continue;
    double q = (double)total / table[i];
    double bits = log2(q);
    totalBits += (bits * table[i]);
  }
  return (long)((totalBits + 7) / 8);
}

// row differences (setRowsDerivative phase 1, fpl_UnitTypes.cpp:302-357) / column differences (setCrossDerivative phase 2, :436-517)
template<class W> static void fplRowDiff(W* d, size_t cols, size_t rows)
{
  for (size_t r = 0; r < rows; r++)
    for (size_t i = cols - 1; i >= 1; i--) d[r * cols + i] = fplSub(d[r * cols + i], d[r * cols + i - 1]);
}
template<class W> static void fplColDiff(W* d, size_t cols, size_t rows)
{
  for (size_t c = 0; c < cols; c++)
    for (size_t r = rows - 1; r >= 1; r--) d[r * cols + c] = fplSub(d[r * cols + c], d[(r - 1) * cols + c]);
}

struct FplBlock { long top, height; };

// fpl_Lerc2Ext.cpp:57-101
static void fplTestBlocks(int width, int height, std::vector<FplBlock>& blocks)
{
  size_t size = (size_t)width * height;
  const int target = 8 * 1024;
  double t = round((double)size / target);
  int count = (int)round(sqrt(t + 1));
  int blockHeight = target / width;
  if (blockHeight < 4) blockHeight = 4;
  while ((count * blockHeight > height) && (count > 1)) count--;
  float topMargin = (float)((height - count * blockHeight) / (2.0 * count));
  float delta = 2.0f * topMargin + blockHeight;
  for (int i = 0; i < count; i++)
  {
    FplBlock tb;
    tb.top = (long)(topMargin + delta * i);
    tb.height = blockHeight;
    if (tb.top < 0) tb.top = 0;
    if (tb.top + tb.height > height) tb.height = height - tb.top;
    if (tb.height > 0) blocks.push_back(tb);
  }
}

// fpl_Lerc2Ext.cpp:167-232 (test_first_byte_delta is always on)
static size_t fplTestBlocksSize(const std::vector<FplBlock>& blocks, size_t unit, const u8* data, long width)
{
  size_t ret = 0;
  std::vector<u8> plane;
  for (const FplBlock& tb : blocks)
  {
    size_t start = unit * tb.top * width, length = (size_t)tb.height * width;
    plane.resize(length);
    for (size_t byte = 0; byte < unit; byte++)
    {
      for (size_t i = 0; i < length; i++) plane[i] = data[start + byte + i * unit];
      size_t plain = (size_t)fplEntropy(plane.data(), length);
      // setDerivativePrime (:81-95): only the sampled bytes are differenced
      for (long i = kFplPrime * (((long)length - 1) / kFplPrime); i >= 1; i -= kFplPrime) plane[i] = (u8)(plane[i] - plane[i - 1]);
      size_t prime = (size_t)fplEntropy(plane.data(), length);
      ret += std::min(plain, prime);
    }
  }
  return ret;
}

// fpl_Lerc2Ext.cpp:237-322
static int fplBestLevel(const u8* p, size_t size, int maxOrder)
{
  if (maxOrder == 0) return 0;
  std::vector<std::pair<size_t, int> > snippets;
  const unsigned target = 1024 * 8;
  double t = round((double)size / target);
  int count = (int)round(sqrt(t + 1));
  while (count * target > size && (count > 0)) count--;
  if (count > 0)    // (the reference divides by zero otherwise and likewise ends up without snippets)
  {
    float topMargin = (float)(((int)size - count * target) / (2.0 * count));
    float delta = 2.0f * topMargin + target;
    for (int i = 0; i < count; i++)
    {
      long start = (long)(topMargin + delta * i);
      int len = (int)target;
      if (start < 0) start = 0;
      if (start + len > (int)size) len = (int)size - start;
      if (len > 0) snippets.push_back(std::make_pair((size_t)start, len));
    }
  }
  std::vector<u8> copy(p, p + size);
  size_t best = 0;
  int ret = 0;
  for (int l = 0; l <= maxOrder; l++)
  {
    if (l > 0)
      for (const auto& sn : snippets)
        for (int i = (int)sn.first + sn.second - 1; i >= (int)sn.first + l; i--) copy[i] = (u8)(copy[i] - copy[i - 1]);
    size_t comp = 0;
    for (const auto& sn : snippets) comp += (size_t)fplEntropy(copy.data() + sn.first, sn.second);
    if (comp < best || l == 0) { best = comp; ret = l; }
    else break;
  }
  return ret;
}

// fpl_EsriHuffman.cpp:164-236 (the limit only shortens the reference's loop; the caller's comparisons decide the same)
static long fplPackBitsSize(const u8* ptr, size_t size, long limit)
{
  long curr = 0;
  int literalCount = 0, literalPos = -1;
  for (size_t i = 0; i <= size; )
  {
    int b = (i == size) ? -1 : ptr[i];
    if (curr > limit) return -1;
    int repeat = 0;
    while (i < size - 1 && b == ptr[i + 1] && repeat < 128) { i++; repeat++; }
    i++;
    if (repeat == 0 && b >= 0)
    {
      if (literalPos < 0) { literalPos = (int)curr; curr++; }
      curr++;
      literalCount++;
      if (literalCount == 128) { literalCount = 0; literalPos = -1; }
    }
    else
    {
      if (literalCount > 0) { literalPos = -1; literalCount = 0; }
      if (repeat > 0) curr += 2;
    }
  }
  return curr;
}

// fpl_EsriHuffman.cpp:79-161
static long fplPackBitsEncode(const u8* ptr, size_t size, u8* out)
{
  int literalCount = 0, curr = 0, literalPos = -1;
  for (size_t i = 0; i <= size; )
  {
    int b = (i == size) ? -1 : ptr[i];
    int repeat = 0;
    while (i < size - 1 && b == ptr[i + 1] && repeat < 128) { i++; repeat++; }
    i++;
    if (repeat == 0 && b >= 0)
    {
      if (literalPos < 0) { literalPos = curr; curr++; }
      out[curr++] = (u8)b;
      literalCount++;
      if (literalCount == 128) { out[literalPos] = (u8)(literalCount - 1); literalCount = 0; literalPos = -1; }
    }
    else
    {
      if (literalCount > 0) { out[literalPos] = (u8)(literalCount - 1); literalPos = -1; literalCount = 0; }
      if (repeat > 0) { out[curr++] = (u8)(127 + repeat); out[curr++] = (u8)b; }
    }
  }
  return curr;
}

// fpl_EsriHuffman.cpp:37-77
static bool fplPackBitsDecode(const u8* ptr, size_t size, size_t expected, u8* out)
{
  size_t curr = 0;
  for (size_t i = 0; i < size; )
  {
    int b = ptr[i++];
    if (b <= 127)
    {
      if (curr + b >= expected || i + b + 1 > size) return false;
      memcpy(&out[curr], &ptr[i], b + 1);
      curr += b + 1; i += b + 1;
    }
    else
    {
      if (curr + b - 127 >= expected || i >= size) return false;
      memset(&out[curr], ptr[i], b - 127 + 1);
      curr += b - 127 + 1; i++;
    }
  }
  return curr == expected;
}

// fpl_EsriHuffman.cpp:306-437; the read-ahead word behind the codes is left uninitialised by the reference, 0 here
static bool fplCompressPlane(const u8* in, size_t len, std::vector<u8>& out)
{
  std::vector<int> histo(256, 0);
  for (size_t i = 0; i < len; i++) histo[in[i]]++;
  int distinct = 0;
  for (int i = 0; i < 256; i++) if (histo[i] > 0) distinct++;
  if (distinct < 2)
  {
    out.assign(6, 0);
    out[0] = 1; out[1] = in[0];
    uint32_t n = (uint32_t)len;
    memcpy(&out[2], &n, 4);
    return true;
  }
  std::vector<HCode> codes;
  int numBytes = 0;
  if (!huffBuild(histo, codes) || !huffCompressedBytes(codes, histo, numBytes) || numBytes <= 0) return false;
  long limit = std::min(numBytes, (int)len);
  long rle = fplPackBitsSize(in, len, limit);
  if (rle > 0 && rle < numBytes && rle < (long)len)
  {
    out.assign((size_t)rle + 1, 0);
    out[0] = 3;
    fplPackBitsEncode(in, len, &out[1]);
    return true;
  }
  if (numBytes >= (int)len)
  {
    out.resize(len + 1);
    out[0] = 2;
    memcpy(&out[1], in, len);
    return true;
  }
  out.assign((size_t)numBytes + 1 + 8, 0);
  out[0] = 0;
  Writer w{ &out[1] };
  if (!huffWriteTable(w, codes)) return false;
  HuffSink sink{ w.p, 0 };
  for (size_t m = 0; m < len; m++)
  {
    const HCode& c = codes[in[m]];
    if (c.first <= 0) return false;
    sink.push(c.second, c.first);
  }
  size_t used = (size_t)(sink.p - out.data()) + 4 * ((sink.bitPos > 0 ? 1 : 0) + 1);
  out.resize(used);
  return true;
}

// fpl_EsriHuffman.cpp:439-560
static bool fplExtractPlane(const u8* in, size_t inCount, size_t expected, u8* out)
{
  if (inCount < 1) return false;
  if (in[0] == 1)
  {
    if (inCount < 6) return false;
    uint32_t n;
    memcpy(&n, in + 2, 4);
    if (n != expected) return false;
    memset(out, in[1], expected);
    return true;
  }
  if (in[0] == 2)
  {
    if (inCount < expected + 1) return false;
    memcpy(out, in + 1, expected);
    return true;
  }
  if (in[0] == 3) return fplPackBitsDecode(in + 1, inCount - 1, expected, out);
  if (in[0] != 0) return false;
  Reader r{ in + 1, inCount - 1 };
  std::vector<HCode> table;
  if (!huffReadTable(r, table, 5)) return false;
  HuffDecoder dec;
  if (!dec.init(table)) return false;
  const u8* p = r.p;
  size_t left = r.left;
  int bitPos = 0;
  for (size_t m = 0; m < expected; m++)
  {
    int sym = 0;
    if (!dec.next(p, left, bitPos, sym)) return false;
    out[m] = (u8)sym;
  }
  return true;
}

struct FplPlane { u8 byteIndex, level; std::vector<u8> bytes; };

// what the reference's LosslessFPCompression object holds between ComputeHuffmanCodesFlt and EncodeHuffmanFlt
struct FplState
{
  std::vector<FplPlane> planes;
  u8 predictor = 0;
  int compressedLength() const    // fpl_Lerc2Ext.cpp:391-403
  {
    int ret = 1;
    for (const FplPlane& b : planes) ret += (int)b.bytes.size() + 6;
    return ret;
  }
};

// fpl_Lerc2Ext.cpp:456-606
template<class W>
static bool fplComputeSlice(const W* input, int cols, int rows, FplState& st)
{
  const size_t unit = sizeof(W), size = (size_t)cols * rows;
  std::vector<W> values(input, input + size);
  if (unit == 4) for (size_t i = 0; i < size; i++) values[i] = (W)fplFwd32((uint32_t)values[i]);
  size_t stats[3] = { 0, 0, 0 };
  {
    std::vector<W> copy(values);    // selectInitialLinearOrCrossDelta (:337-389)
    std::vector<FplBlock> blocks;
    fplTestBlocks(cols, rows, blocks);
    stats[0] = fplTestBlocksSize(blocks, unit, (const u8*)copy.data(), cols);
    fplRowDiff(copy.data(), cols, rows);
    stats[1] = fplTestBlocksSize(blocks, unit, (const u8*)copy.data(), cols);
    fplColDiff(copy.data(), cols, rows);
    stats[2] = fplTestBlocksSize(blocks, unit, (const u8*)copy.data(), cols);
  }
  int predictor = 0;
  for (int i = 1; i < 3; i++) if (stats[i] < stats[predictor]) predictor = i;
  if (predictor >= 1) fplRowDiff(values.data(), cols, rows);
  if (predictor == 2) fplColDiff(values.data(), cols, rows);
  const int maxDelta = kFplMaxDelta - predictor;    // Predictor::getMaxByteDelta
  std::vector<u8> plane(size);
  const u8* bytes = (const u8*)values.data();
  for (size_t byte = 0; byte < unit; byte++)
  {
    for (size_t i = 0; i < size; i++) plane[i] = bytes[i * unit + byte];
    const int level = fplBestLevel(plane.data(), size, maxDelta);
    for (int l = 1; l <= level; l++)    // setDerivative (:97-110)
      for (int i = (int)size - 1; i >= l; i--) plane[i] = (u8)(plane[i] - plane[i - 1]);
    FplPlane out;
    out.byteIndex = (u8)byte; out.level = (u8)level;
    if (!fplCompressPlane(plane.data(), size, out.bytes)) return false;
    st.predictor = (u8)predictor;
    st.planes.push_back(out);
  }
  return true;
}

// fpl_Lerc2Ext.cpp:423-452: nDepth > 1 is coded as an (nDepth x nPixels) raster, and -- unlike the nDepth == 1 entry --
// does not drop planes a previous band computed but never wrote
template<class T>
static bool fplCompute(const T* data, int nCols, int nRows, int nDepth, FplState& st)
{
  typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type W;
  if (nDepth == 1) { st.planes.clear(); return fplComputeSlice((const W*)data, nCols, nRows, st); }
  return fplComputeSlice((const W*)data, nDepth, nCols * nRows, st);
}

// fpl_Lerc2Ext.cpp:405-421
static void fplWrite(FplState& st, Writer& w)
{
  w.byte(st.predictor);
  for (const FplPlane& b : st.planes)
  {
    w.byte(b.byteIndex); w.byte(b.level);
    uint32_t n = (uint32_t)b.bytes.size();
    w.put(&n, 4);
    w.put(b.bytes.data(), b.bytes.size());
  }
  st.planes.clear();
}

// fpl_Lerc2Ext.cpp:723-866
template<class T>
static bool fplDecode(Reader& r, T* data, int nCols, int nRows, int nDepth)
{
  typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type W;
  const size_t unit = sizeof(W);
  const size_t cols = (nDepth == 1) ? nCols : nDepth, rows = (nDepth == 1) ? nRows : (size_t)nCols * nRows, size = cols * rows;
  u8 pred = 0;
  if (!r.get(&pred, 1) || pred > 2) return false;
  std::vector<W> values(size, 0);
  u8* bytes = (u8*)values.data();
  std::vector<u8> plane(size);
  for (size_t b = 0; b < unit; b++)
  {
    u8 byteIndex = 0, level = 0;
    uint32_t csize = 0;
    if (r.left < 6) return false;
    r.get(&byteIndex, 1); r.get(&level, 1); r.get(&csize, 4);
    if (byteIndex >= unit || level > kFplMaxDelta || r.left < csize) return false;
    if (!fplExtractPlane(r.p, csize, size, plane.data())) return false;
    r.skip(csize);
    for (int l = level; l > 0; l--)    // restoreSequence (:128-165)
      for (size_t i = l; i < size; i++) plane[i] = (u8)(plane[i] + plane[i - 1]);
    for (size_t i = 0; i < size; i++) bytes[i * unit + byteIndex] = plane[i];
  }
  if (pred == 2)    // restoreCrossBytes (fpl_UnitTypes.cpp:775-849)
    for (size_t c = 0; c < cols; c++)
      for (size_t i = 1; i < rows; i++) values[i * cols + c] = fplAdd(values[i * cols + c], values[(i - 1) * cols + c]);
  if (pred >= 1)    // ... and restoreBlockSequence (:626-697)
    for (size_t i = 0; i < rows; i++)
      for (size_t c = 1; c < cols; c++) values[i * cols + c] = fplAdd(values[i * cols + c], values[i * cols + c - 1]);
  if (unit == 4) for (size_t i = 0; i < size; i++) values[i] = (W)fplBack32((uint32_t)values[i]);
  memcpy(data, values.data(), size * unit);
  return true;
}

// ---------------------------------------------------------------------------------------------
// per-block helpers -- Lerc2.h:337-353 (ComputeMaxVal / NeedToQuantize), :357-376 (Quantize),
// :457-515 (ReduceDataType), :528-542 (GetDataTypeUsed), :546-681 (variable-type offset I/O)
// ---------------------------------------------------------------------------------------------
static double maxValOf(double zMin, double zMax, double maxZErr)
{
  double fac = 1 / (2 * maxZErr);
  return (zMax - zMin) * fac;
}

template<class Z>
static int reduceType(Z z, int dt, int& dtRed)
{
  u8 b = (z >= 0 && z <= 255) ? (u8)z : 0;
  switch (dt)
  {
    case DT_SHORT:
    {
      signed char c = (z >= (double)-128 && z <= 127) ? (signed char)z : 0;
      int tc = (Z)c == z ? 2 : (Z)b == z ? 1 : 0;
      dtRed = dt - tc;
      return tc;
    }
    case DT_USHORT:
    {
      int tc = (Z)b == z ? 1 : 0;
      dtRed = dt - 2 * tc;
      return tc;
    }
    case DT_INT:
    {
      short s = (z >= (double)SHRT_MIN && z <= SHRT_MAX) ? (short)z : 0;
      unsigned short us = (z >= 0 && z <= USHRT_MAX) ? (unsigned short)z : 0;
      int tc = (Z)b == z ? 3 : (Z)s == z ? 2 : (Z)us == z ? 1 : 0;
      dtRed = dt - tc;
      return tc;
    }
    case DT_UINT:
    {
      unsigned short us = (z >= 0 && z <= USHRT_MAX) ? (unsigned short)z : 0;
      int tc = (Z)b == z ? 2 : (Z)us == z ? 1 : 0;
      dtRed = dt - 2 * tc;
      return tc;
    }
    case DT_FLOAT:
    {
      short s = (z >= (float)SHRT_MIN && z <= SHRT_MAX) ? (short)z : 0;
      int tc = (Z)b == z ? 2 : (Z)s == z ? 1 : 0;
      dtRed = tc == 0 ? dt : (tc == 1 ? DT_SHORT : DT_BYTE);
      return tc;
    }
    case DT_DOUBLE:
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

static int typeUsed(int dt, int tc)
{
  int r;
  switch (dt)
  {
    case DT_SHORT: case DT_INT: r = dt - tc; break;
    case DT_USHORT: case DT_UINT: r = dt - 2 * tc; break;
    case DT_FLOAT: return tc == 0 ? dt : (tc == 1 ? DT_SHORT : DT_BYTE);
    case DT_DOUBLE: r = tc == 0 ? dt : dt - 2 * tc + 1; break;
    default: return dt;
  }
  return (r >= DT_CHAR && r <= DT_DOUBLE) ? r : DT_UNDEF;
}

static bool putTyped(Writer& w, double z, int dt)
{
  switch (dt)
  {
    case DT_CHAR:   { signed char v = (signed char)z; w.put(&v, 1); return true; }
    case DT_BYTE:   { u8 v = (u8)z; w.put(&v, 1); return true; }
    case DT_SHORT:  { short v = (short)z; w.put(&v, 2); return true; }
    case DT_USHORT: { unsigned short v = (unsigned short)z; w.put(&v, 2); return true; }
    case DT_INT:    { int v = (int)z; w.put(&v, 4); return true; }
    case DT_UINT:   { unsigned v = (unsigned)z; w.put(&v, 4); return true; }
    case DT_FLOAT:  { float v = (float)z; w.put(&v, 4); return true; }
    case DT_DOUBLE: { w.put(&z, 8); return true; }
    default: return false;
  }
}

static double getTyped(const u8* p, int dt)
{
  switch (dt)
  {
    case DT_CHAR:   { signed char v; memcpy(&v, p, 1); return v; }
    case DT_BYTE:   { u8 v; memcpy(&v, p, 1); return v; }
    case DT_SHORT:  { short v; memcpy(&v, p, 2); return v; }
    case DT_USHORT: { unsigned short v; memcpy(&v, p, 2); return v; }
    case DT_INT:    { int v; memcpy(&v, p, 4); return v; }
    case DT_UINT:   { unsigned v; memcpy(&v, p, 4); return v; }
    case DT_FLOAT:  { float v; memcpy(&v, p, 4); return v; }
    case DT_DOUBLE: { double v; memcpy(&v, p, 8); return v; }
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------------------
// one band: the Lerc2 object of the reference (Lerc2.h:71-290), flattened
// ---------------------------------------------------------------------------------------------
struct Band
{
  Header hd;
  Mask mask;                          // persists across bands ("use previous mask", Lerc2.cpp:1002)
  bool encodeMask = true, oneSweep = false, minMaxSet = false;
  int imageMode = IEM_TILING;
  unsigned maxQ = 0;
  std::vector<double> zMinVec, zMaxVec;
  std::vector<HCode> huffCodes;
  FplState fpl;                       // lossless float planes between plan() and emit() (m_lfpc, Lerc2.h)

  // Lerc2.cpp:85-114
  bool setDims(int nDepth, int nCols, int nRows, const u8* maskBits)
  {
    mask.resize(nCols, nRows);
    if (maskBits)
    {
      memcpy(mask.bits.data(), maskBits, mask.nBytes());
      int64_t nv = mask.countValid();
      if (nv < 0 || nv > INT_MAX) return false;
      hd.numValid = (int)nv;
    }
    else { hd.numValid = nCols * nRows; mask.fill(true); }
    hd.nDepth = nDepth; hd.nCols = nCols; hd.nRows = nRows;
    return true;
  }
  bool allValid() const { return hd.numValid == hd.nCols * hd.nRows; }

  template<class T> unsigned plan(const T* data, double maxZErr, bool encMask);
  template<class T> bool emit(const T* data, Writer& w);
  template<class T> bool decode(Reader& r, T* data, u8* maskBitsOut);
  bool ranges(const u8* p, size_t n, double* mins, double* maxs);

  template<class T> bool minMaxRanges(const T* data);
  template<class T> bool tryRaiseMaxZErr(const T* data, double& maxZErr) const;
  template<class T> bool tryBitPlanes(const T* data, double eps, double& newMaxZErr) const;
  template<class T> bool tilesPass(const T* data, Writer* w, int64_t& nBytes) const;
  template<class T> void blockStats(const T* data, int i0, int i1, int j0, int j1, int iDepth, T* buf, T& zMin, T& zMax,
    int& n, bool& tryLut) const;
  template<class Z> bool needQuant(int n, Z zMin, Z zMax) const;
  template<class Z> void quantize(const Z* buf, int n, Z zMin, std::vector<unsigned>& q) const;
  template<class Z> int blockBytes(int n, Z zMin, Z zMax, int dtZ, bool tryLut, int& mode, const std::vector<QIdx>& sorted) const;
  template<class Z> bool writeBlock(const Z* buf, int n, Writer& w, int& nWritten, int j0, Z zMin, Z zMax, int dtZ,
    bool diff, const std::vector<unsigned>& q, int mode, const std::vector<QIdx>& sorted) const;
  template<class T> bool readTiles(Reader& r, T* data) const;
  template<class T> bool readBlock(Reader& r, T* data, int i0, int i1, int j0, int j1, int iDepth, std::vector<unsigned>& buf) const;
  template<class T> void huffHistos(const T* data, std::vector<int>& histo, std::vector<int>& dHisto) const;
  template<class T> void huffChoose(const T* data, int& nBytes, int& mode, std::vector<HCode>& codes) const;
  template<class T> bool huffEncode(const T* data, Writer& w) const;
  template<class T> bool huffDecode(Reader& r, T* data) const;
  template<class T> bool fillConst(T* data) const;
  bool readMask(Reader& r);
};

// Lerc2.cpp:1404-1470 -- per-depth min/max over valid pixels
template<class T> bool Band::minMaxRanges(const T* data)
{
  if (hd.numValid == 0) return false;
  const int nD = hd.nDepth;
  std::vector<T> lo(nD, 0), hi(nD, 0);
  bool init = false;
  for (int64_t k = 0, n = (int64_t)hd.nRows * hd.nCols; k < n; k++)
  {
    if (!mask.valid(k)) continue;
    const T* px = data + k * nD;
    if (!init) { for (int m = 0; m < nD; m++) lo[m] = hi[m] = px[m]; init = true; continue; }
    for (int m = 0; m < nD; m++)
    {
      T v = px[m];
      if (v < lo[m]) lo[m] = v; else if (v > hi[m]) hi[m] = v;
    }
  }
  zMinVec.resize(nD); zMaxVec.resize(nD);
  if (init) for (int m = 0; m < nD; m++) { zMinVec[m] = lo[m]; zMaxVec[m] = hi[m]; }
  return init;
}

// Lerc2.cpp:1233-1339 -- is the float data decimal-rounded so that a larger error bound is free?
template<class T> bool Band::tryRaiseMaxZErr(const T* data, double& maxZErr) const
{
  if (hd.dt < DT_FLOAT || hd.numValid == 0) return false;
  static const double errCand[] = { 1, 0.5, 0.1, 0.05, 0.01, 0.005, 0.001, 0.0005, 0.0001 };
  static const int facCand[] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
  std::vector<double> rnd, zErr;
  std::vector<int> fac;
  for (int i = 0; i < 9; i++)
    if (errCand[i] / 2 > maxZErr) { zErr.push_back(errCand[i] / 2); fac.push_back(facCand[i]); rnd.push_back(0); }
  if (zErr.empty()) return false;

  const int nD = hd.nDepth;
  for (int i = 0; i < hd.nRows; i++)
  {
    size_t nCand = zErr.size();
    for (int j = 0; j < hd.nCols; j++)
    {
      int64_t k = (int64_t)i * hd.nCols + j;
      if (!mask.valid(k)) continue;
      for (int m = 0; m < nD; m++)
      {
        double x = data[k * nD + m];
        for (size_t c = 0; c < nCand; c++)
        {
          double z = x * fac[c];
          if (z == (int)z) break;
          double d = fabs(floor(z + 0.5) - z);
          rnd[c] = std::max(rnd[c], d);
        }
      }
    }
    // prune after every row (Lerc2.cpp:1322-1339)
    if (maxZErr <= 0) return false;
    for (int c = (int)zErr.size() - 1; c >= 0; c--)
      if (rnd[c] / fac[c] > maxZErr / 2) { rnd.erase(rnd.begin() + c); zErr.erase(zErr.begin() + c); fac.erase(fac.begin() + c); }
    if (zErr.empty()) return false;
  }
  for (size_t c = 0; c < zErr.size(); c++)
    if (rnd[c] / fac[c] <= maxZErr / 2) { maxZErr = zErr[c]; return true; }
  return false;
}

// Lerc2.cpp:1071-1229 -- integer "bit plane" mode (maxZErr == 777 cheat code / negative maxZErr)
template<class T> bool Band::tryBitPlanes(const T* data, double eps, double& newMaxZErr) const
{
  newMaxZErr = 0;
  if (eps <= 0) return false;
  const int nD = hd.nDepth, nBits = 8 * (int)dtSize(hd.dt), minCnt = 5000;
  if (hd.numValid < minCnt) return false;
  if (hd.dt >= DT_FLOAT) return false;
  std::vector<int> cntDiff((size_t)nD * nBits, 0);
  int cnt = 0;
  const bool isSigned = (hd.dt == DT_CHAR || hd.dt == DT_SHORT || hd.dt == DT_INT);
  auto tally = [&](int* c, T a, T b)
  {
    unsigned x = isSigned ? (unsigned)((int)a ^ (int)b) : ((unsigned)a ^ (unsigned)b);
    if (isSigned) { int v = (int)x; c[0] += v & 1; for (int s = 1; s < nBits; s++) c[s] += (v >>= 1) & 1; }
    else { c[0] += x & 1; for (int s = 1; s < nBits; s++) c[s] += (x >>= 1) & 1; }
  };
  if (nD == 1 && allValid())
  {
    for (int i = 0; i < hd.nRows - 1; i++)
      for (int j = 0; j < hd.nCols - 1; j++)
      {
        int64_t k = (int64_t)i * hd.nCols + j;
        tally(&cntDiff[0], data[k], data[k + 1]); cnt++;
        tally(&cntDiff[0], data[k], data[k + hd.nCols]); cnt++;
      }
  }
  else
  {
    for (int i = 0; i < hd.nRows; i++)
      for (int j = 0; j < hd.nCols; j++)
      {
        int64_t k = (int64_t)i * hd.nCols + j, m0 = k * nD;
        if (!mask.valid(k)) continue;
        if (j < hd.nCols - 1 && mask.valid(k + 1))
        {
          for (int m = 0; m < nD; m++) tally(&cntDiff[(size_t)m * nBits], data[m0 + m], data[m0 + m + nD]);
          cnt++;
        }
        if (i < hd.nRows - 1 && mask.valid(k + hd.nCols))
        {
          for (int m = 0; m < nD; m++) tally(&cntDiff[(size_t)m * nBits], data[m0 + m], data[m0 + m + (int64_t)nD * hd.nCols]);
          cnt++;
        }
      }
  }
  if (cnt < minCnt) return false;
  int nCut = 0, lastKept = 0;
  for (int s = nBits - 1; s >= 0; s--)
  {
    bool crit = true;
    for (int m = 0; m < nD; m++)
    {
      double x = cntDiff[(size_t)m * nBits + s], n = cnt;
      if (fabs(1 - 2 * (x / n)) >= eps) crit = false;
    }
    if (crit && nCut < 2)
    {
      if (nCut == 0) lastKept = s;
      if (nCut == 1 && s < lastKept - 1) { lastKept = s; nCut = 0; }
      nCut++;
    }
  }
  lastKept = std::max(0, lastKept);
  newMaxZErr = (1 << lastKept) >> 1;
  return true;
}

template<class Z> bool Band::needQuant(int n, Z zMin, Z zMax) const
{
  if (n == 0 || hd.maxZErr == 0) return false;
  double mv = maxValOf((double)zMin, (double)zMax, hd.maxZErr);
  return !(mv > maxQ || (unsigned)(mv + 0.5) == 0);
}

template<class Z> void Band::quantize(const Z* buf, int n, Z zMin, std::vector<unsigned>& q) const
{
  q.resize(n);
  if (hd.dt < DT_FLOAT && hd.maxZErr == 0.5)
  {
    // integer lossless: plain difference in the promoted type (Lerc2.h:362-366)
    for (int i = 0; i < n; i++)
    {
      if (std::is_same<Z, unsigned int>::value) q[i] = (unsigned)buf[i] - (unsigned)zMin;
      else q[i] = (unsigned)((int64_t)buf[i] - (int64_t)zMin);
    }
  }
  else
  {
    double scale = 1 / (2 * hd.maxZErr), z0 = (double)zMin;
    for (int i = 0; i < n; i++) q[i] = (unsigned)(((double)buf[i] - z0) * scale + 0.5);
  }
}

// Lerc2.cpp:1717-1799
template<class T> void Band::blockStats(const T* data, int i0, int i1, int j0, int j1, int iDepth, T* buf, T& zMin,
  T& zMax, int& n, bool& tryLut) const
{
  zMin = zMax = 0;
  tryLut = false;
  T prev = 0;
  int cnt = 0, same = 0;
  const int nD = hd.nDepth;
  if (allValid())
  {
    zMin = zMax = data[((int64_t)i0 * hd.nCols + j0) * nD + iDepth];
    for (int i = i0; i < i1; i++)
      for (int j = j0; j < j1; j++)
      {
        T v = data[((int64_t)i * hd.nCols + j) * nD + iDepth];
        buf[cnt++] = v;
        if (v < zMin) zMin = v; else if (v > zMax) zMax = v;
        if (v == prev) same++;
        prev = v;
      }
  }
  else
  {
    for (int i = i0; i < i1; i++)
      for (int j = j0; j < j1; j++)
      {
        int64_t k = (int64_t)i * hd.nCols + j;
        if (!mask.valid(k)) continue;
        T v = data[k * nD + iDepth];
        buf[cnt] = v;
        if (cnt > 0)
        {
          if (v < zMin) zMin = v; else if (v > zMax) zMax = v;
          if (v == prev) same++;
        }
        else zMin = zMax = v;
        prev = v;
        cnt++;
      }
  }
  if (cnt > 4) tryLut = (zMax > zMin + 3 * hd.maxZErr) && (2 * same > cnt);
  n = cnt;
}

// Lerc2.h:416-453
template<class Z> int Band::blockBytes(int n, Z zMin, Z zMax, int dtZ, bool tryLut, int& mode,
  const std::vector<QIdx>& sorted) const
{
  mode = BEM_RAW;
  if (n == 0 || (zMin == 0 && zMax == 0)) return 1;
  double mv = 0, e = hd.maxZErr;
  int raw = (int)(1 + n * sizeof(Z));
  if ((e == 0 && zMax > zMin) || (e > 0 && (mv = maxValOf((double)zMin, (double)zMax, e)) > maxQ))
    return raw;
  int dtRed;
  reduceType(zMin, dtZ, dtRed);
  int nb = 1 + (int)dtSize(dtRed);
  unsigned maxElem = (unsigned)(mv + 0.5);
  if (maxElem > 0)
    nb += !tryLut ? (int)sizeSimple((unsigned)n, maxElem) : (int)sizeLut(sorted, tryLut);
  if (nb < raw) mode = (!tryLut || maxElem == 0) ? BEM_SIMPLE : BEM_LUT;
  else nb = raw;
  return nb;
}

// Lerc2.cpp:1949-2021
template<class Z> bool Band::writeBlock(const Z* buf, int n, Writer& w, int& nWritten, int j0, Z zMin, Z zMax,
  int dtZ, bool diff, const std::vector<unsigned>& q, int mode, const std::vector<QIdx>& sorted) const
{
  u8* start = w.p;
  u8 flag = (u8)(((j0 >> 3) & 15) << 2);
  if (hd.version >= 5) flag = diff ? (flag | 4) : (flag & (7 << 3));
  if (n == 0 || (zMin == 0 && zMax == 0)) { w.byte(flag | 2); nWritten = 1; return true; }
  if (mode == BEM_RAW)
  {
    if (diff) return false;
    w.byte(flag);
    w.put(buf, n * sizeof(Z));
  }
  else
  {
    double mv = hd.maxZErr > 0 ? maxValOf((double)zMin, (double)zMax, hd.maxZErr) : 0;
    unsigned maxElem = (unsigned)(mv + 0.5);
    flag |= (maxElem == 0) ? 3 : 1;
    int dtRed;
    int tc = reduceType(zMin, dtZ, dtRed);
    flag |= (u8)(tc << 6);
    w.byte(flag);
    if (!putTyped(w, (double)zMin, dtRed)) return false;
    if (maxElem > 0)
    {
      if ((int)q.size() != n) return false;
      if (mode == BEM_SIMPLE) { if (!encodeSimple(w, q, hd.version)) return false; }
      else if (mode == BEM_LUT) { if (!encodeLut(w, sorted, hd.version)) return false; }
      else return false;
    }
  }
  nWritten = (int)(w.p - start);
  return true;
}

// Lerc2.cpp:1803-1874 (integer slice difference; the float variant is unreachable because diff
// encoding is only tried for integer lossless, Lerc2.cpp:1495)
template<class T>
static bool diffSliceInt(const T* cur, const T* prev, int n, bool checkOverflow, double maxZErr, std::vector<int>& d,
  int& zMin, int& zMax, bool& tryLut)
{
  if (n <= 0) return false;
  d.resize(n);
  int prevVal = 0, same = 0;
  bool overflow = false;
  for (int i = 0; i < n; i++)
  {
    int v;
    if (!checkOverflow) v = (int)cur[i] - (int)prev[i];
    else
    {
      double z = (double)cur[i] - (double)prev[i];
      if (z < -2147483648.0 || z > 2147483647.0) { overflow = true; v = (z < 0) ? INT_MIN : INT_MIN; }
      else v = (int)z;
    }
    d[i] = v;
    if (i == 0) zMin = zMax = v;
    if (v < zMin) zMin = v; else if (v > zMax) zMax = v;
    if (v == prevVal) same++;
    prevVal = v;
  }
  if (overflow) return false;
  if (n > 4) tryLut = (zMax > zMin + 3 * maxZErr) && (2 * same > n);
  return true;
}

static void sortQuant(const std::vector<unsigned>& q, std::vector<QIdx>& s)
{
  s.resize(q.size());
  for (size_t i = 0; i < q.size(); i++) s[i] = QIdx(q[i], (unsigned)i);
  std::sort(s.begin(), s.end(), [](const QIdx& a, const QIdx& b) { return a.first < b.first; });
}

// Lerc2.cpp:1474-1668 -- the block loop; w == nullptr is the size-only dry run
template<class T> bool Band::tilesPass(const T* data, Writer* w, int64_t& nBytes) const
{
  nBytes = 0;
  const int mb = hd.mbSize, nD = hd.nDepth;
  std::vector<T> bufVec((size_t)mb * mb, 0), prevVec;
  std::vector<int> diffVec;
  std::vector<unsigned> q, qDiff;
  std::vector<QIdx> sorted, sortedDiff;
  T* buf = bufVec.data();

  const bool intLossless = (hd.dt < DT_FLOAT) && (hd.maxZErr == 0.5);
  const bool tryDiff = (hd.version >= 5) && (nD > 1) && intLossless;
  const bool checkOverflow = (hd.dt == DT_INT || hd.dt == DT_UINT) && (hd.zMax - hd.zMin >= 0x7FFFFFFF);    // Lerc2.h:312-315
  if (tryDiff) prevVec.assign((size_t)mb * mb, 0);

  const int nTV = (hd.nRows + mb - 1) / mb, nTH = (hd.nCols + mb - 1) / mb;
  for (int it = 0; it < nTV; it++)
  {
    int i0 = it * mb, i1 = std::min(hd.nRows, i0 + mb);
    for (int jt = 0; jt < nTH; jt++)
    {
      int j0 = jt * mb, j1 = std::min(hd.nCols, j0 + mb);
      for (int iD = 0; iD < nD; iD++)
      {
        T zMin = 0, zMax = 0;
        int n = 0;
        bool tryLut = false, quantDone = false;
        blockStats(data, i0, i1, j0, j1, iD, buf, zMin, zMax, n, tryLut);

        if (n == 0 && !w) { nBytes += nD; break; }

        if (((w && iD == 0) || tryLut) && needQuant(n, zMin, zMax))
        {
          quantize(buf, n, zMin, q);
          quantDone = true;
          if (tryLut) sortQuant(q, sorted);
        }
        int mode = BEM_RAW, modeDiff = BEM_RAW;
        int need = blockBytes(n, zMin, zMax, hd.dt, tryLut, mode, sorted);
        int needDiff = need + 1;
        int zMinD = 0, zMaxD = 0;
        bool quantDoneDiff = false, tryLutD = false;

        if (tryDiff && iD > 0 && n > 0)
        {
          if (diffSliceInt(buf, prevVec.data(), n, checkOverflow, hd.maxZErr, diffVec, zMinD, zMaxD, tryLutD))
          {
            if (tryLutD && needQuant(n, (double)zMinD, (double)zMaxD))
            {
              quantize(diffVec.data(), n, zMinD, qDiff);
              quantDoneDiff = true;
              sortQuant(qDiff, sortedDiff);
            }
            int nb = blockBytes(n, zMinD, zMaxD, DT_INT, tryLutD, modeDiff, sortedDiff);
            if (nb > 0) needDiff = nb;
          }
        }
        nBytes += std::min(need, needDiff);

        if (tryDiff && iD < nD - 1 && n > 0)
        {
          if (iD == 0) prevVec.resize(n);
          std::copy(bufVec.begin(), bufVec.begin() + n, prevVec.begin());
        }

        if (w)
        {
          int wrote = 0;
          bool ok;
          if (iD == 0 || need <= needDiff)
          {
            if (!quantDone && needQuant(n, zMin, zMax)) quantize(buf, n, zMin, q);
            ok = writeBlock(buf, n, *w, wrote, j0, zMin, zMax, hd.dt, false, q, mode, sorted);
          }
          else
          {
            if (!quantDoneDiff && needQuant(n, (double)zMinD, (double)zMaxD)) quantize(diffVec.data(), n, zMinD, qDiff);
            ok = writeBlock(diffVec.data(), n, *w, wrote, j0, zMinD, zMaxD, DT_INT, true, qDiff, modeDiff, sortedDiff);
          }
          if (!ok || wrote != std::min(need, needDiff)) return false;
        }
      }
    }
  }
  return true;
}

// Lerc2.cpp:2311-2380
template<class T> void Band::huffHistos(const T* data, std::vector<int>& histo, std::vector<int>& dHisto) const
{
  histo.assign(256, 0);
  dHisto.assign(256, 0);
  const int off = (hd.dt == DT_CHAR) ? 128 : 0, H = hd.nRows, W = hd.nCols, nD = hd.nDepth;
  const bool all = allValid();
  for (int iD = 0; iD < nD; iD++)
  {
    T prev = 0;
    for (int i = 0; i < H; i++)
      for (int j = 0; j < W; j++)
      {
        int64_t k = (int64_t)i * W + j, m = k * nD + iD;
        if (!all && !mask.valid(k)) continue;
        T v = data[m], d = v;
        if (j > 0 && (all || mask.valid(k - 1))) d = (T)(d - prev);
        else if (i > 0 && (all || mask.valid(k - W))) d = (T)(d - data[m - (int64_t)W * nD]);
        else d = (T)(d - prev);
        prev = v;
        histo[off + (int)v]++;
        dHisto[off + (int)d]++;
      }
  }
}

// Lerc2.cpp:2270-2307
template<class T> void Band::huffChoose(const T* data, int& nBytes, int& mode, std::vector<HCode>& codes) const
{
  std::vector<int> h0, h1;
  huffHistos(data, h0, h1);
  std::vector<HCode> t0, t1;
  int n0 = 0, n1 = 0;
  if (hd.version >= 4) { if (!huffBuild(h0, t0) || !huffCompressedBytes(t0, h0, n0)) n0 = 0; }
  if (!huffBuild(h1, t1) || !huffCompressedBytes(t1, h1, n1)) n1 = 0;
  if (n0 > 0 && n1 > 0)
  {
    mode = (n0 <= n1) ? IEM_HUFFMAN : IEM_DELTA_HUFFMAN;
    codes = (n0 <= n1) ? t0 : t1;
    nBytes = std::min(n0, n1);
  }
  else if (n0 == 0 && n1 == 0) { mode = IEM_TILING; codes.clear(); nBytes = 0; }
  else
  {
    mode = (n0 > n1) ? IEM_HUFFMAN : IEM_DELTA_HUFFMAN;
    codes = (n0 > n1) ? t0 : t1;
    nBytes = std::max(n0, n1);
  }
}

// Lerc2.cpp:2384-2468
template<class T> bool Band::huffEncode(const T* data, Writer& w) const
{
  if (!huffWriteTable(w, huffCodes, hd.version)) return false;
  const int off = (hd.dt == DT_CHAR) ? 128 : 0, H = hd.nRows, W = hd.nCols, nD = hd.nDepth;
  const bool all = allValid();
  HuffSink s{ w.p, 0 };
  if (imageMode == IEM_DELTA_HUFFMAN)
  {
    for (int iD = 0; iD < nD; iD++)
    {
      T prev = 0;
      for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++)
        {
          int64_t k = (int64_t)i * W + j, m = k * nD + iD;
          if (!all && !mask.valid(k)) continue;
          T v = data[m], d = v;
          if (j > 0 && (all || mask.valid(k - 1))) d = (T)(d - prev);
          else if (i > 0 && (all || mask.valid(k - W))) d = (T)(d - data[m - (int64_t)W * nD]);
          else d = (T)(d - prev);
          prev = v;
          const HCode& c = huffCodes[off + (int)d];
          if (c.first <= 0) return false;
          s.push(c.second, c.first);
        }
    }
  }
  else if (imageMode == IEM_HUFFMAN)
  {
    for (int64_t k = 0, n = (int64_t)H * W; k < n; k++)
    {
      if (!all && !mask.valid(k)) continue;
      for (int m = 0; m < nD; m++)
      {
        const HCode& c = huffCodes[off + (int)data[k * nD + m]];
        if (c.first <= 0) return false;
        s.push(c.second, c.first);
      }
    }
  }
  else return false;
  w.p = s.p + 4 * ((s.bitPos > 0 ? 1 : 0) + 1);    // one extra word: the decode LUT may read ahead
  return true;
}

// Lerc2.cpp:2472-2606
template<class T> bool Band::huffDecode(Reader& r, T* data) const
{
  std::vector<HCode> table;
  if (!huffReadTable(r, table, hd.version)) return false;
  HuffDecoder dec;
  if (!dec.init(table)) return false;
  const int off = (hd.dt == DT_CHAR) ? 128 : 0, H = hd.nRows, W = hd.nCols, nD = hd.nDepth;
  const bool all = allValid();
  const u8* p0 = r.p;
  const u8* p = p0;
  size_t left = r.left;
  int bitPos = 0;
  if (imageMode == IEM_DELTA_HUFFMAN)
  {
    for (int iD = 0; iD < nD; iD++)
    {
      T prev = 0;
      for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++)
        {
          int64_t k = (int64_t)i * W + j, m = k * nD + iD;
          if (!all && !mask.valid(k)) continue;
          int sym = 0;
          if (!dec.next(p, left, bitPos, sym)) return false;
          T d = (T)(sym - off);
          if (j > 0 && (all || mask.valid(k - 1))) d = (T)(d + prev);
          else if (i > 0 && (all || mask.valid(k - W))) d = (T)(d + data[m - (int64_t)W * nD]);
          else d = (T)(d + prev);
          data[m] = d;
          prev = d;
        }
    }
  }
  else if (imageMode == IEM_HUFFMAN)
  {
    for (int64_t k = 0, n = (int64_t)H * W; k < n; k++)
    {
      if (!all && !mask.valid(k)) continue;
      for (int m = 0; m < nD; m++)
      {
        int sym = 0;
        if (!dec.next(p, left, bitPos, sym)) return false;
        data[k * nD + m] = (T)(sym - off);
      }
    }
  }
  else return false;
  size_t used = (size_t)(p - p0) + 4 * ((bitPos > 0 ? 1 : 0) + 1);
  if (r.left < used) return false;
  r.skip(used);
  return true;
}

// Lerc2.cpp:179-381 -- dry run: fixes maxZError, ranges, image mode, micro block size, blob size
template<class T> unsigned Band::plan(const T* data, double maxZErr, bool encMask)
{
  if (!data) return 0;
  unsigned nHdrMask = headerBytes(hd.version);
  const int numValid = hd.numValid, numTotal = hd.nCols * hd.nRows;
  const bool needMask = numValid > 0 && numValid < numTotal;
  encodeMask = encMask;
  nHdrMask += 4;
  if (needMask && encMask)
  {
    std::vector<u8> rle;
    rleEncode(mask.bits.data(), mask.nBytes(), rle);
    nHdrMask += (unsigned)rle.size();
  }
  hd.dt = DtOf<T>::v;
  if (maxZErr == 777) maxZErr = -0.01;
  if (hd.dt < DT_FLOAT)
  {
    if (maxZErr < 0 && !tryBitPlanes(data, -maxZErr, maxZErr)) maxZErr = 0;
    maxZErr = std::max(0.5, floor(maxZErr));
  }
  else
  {
    if (maxZErr < 0) return 0;
    double raised = maxZErr;
    if (maxZErr > 0 && tryRaiseMaxZErr(data, raised)) maxZErr = raised;
  }
  hd.maxZErr = maxZErr;
  hd.zMin = hd.zMax = 0;
  hd.mbSize = 8;
  hd.blobSize = (int)nHdrMask;
  if (numValid == 0) return nHdrMask;

  maxQ = maxValToQuantize(hd.dt);
  if ((!minMaxSet || hd.nDepth > 1) && !minMaxRanges(data)) return 0;
  hd.zMin = *std::min_element(zMinVec.begin(), zMinVec.end());
  hd.zMax = *std::max_element(zMaxVec.begin(), zMaxVec.end());
  if (hd.zMin == hd.zMax) return nHdrMask;

  const int nD = hd.nDepth;
  if (hd.version >= 4)
  {
    size_t sz = (size_t)hd.blobSize + sizeof(T) * nD * 2;
    if (sz > (size_t)INT_MAX) return 0;
    hd.blobSize = (int)sz;
    if ((int)zMinVec.size() != nD || (int)zMaxVec.size() != nD) return 0;
    if (0 == memcmp(zMinVec.data(), zMaxVec.data(), nD * sizeof(double))) return hd.blobSize;
  }

  int64_t nTiling64 = 0;
  if (!tilesPass<T>(data, nullptr, nTiling64) || nTiling64 > INT_MAX) return 0;
  int nBytesTiling = (int)nTiling64;
  imageMode = IEM_TILING;
  int nBytesData = nBytesTiling, nBytesHuff = 0;

  if (hd.tryHuffmanInt())
  {
    int hm = IEM_TILING;
    huffChoose(data, nBytesHuff, hm, huffCodes);
    if (nBytesHuff < 0) nBytesHuff = INT_MAX;
    if (!huffCodes.empty() && nBytesHuff < nBytesTiling) { imageMode = hm; nBytesData = nBytesHuff; }
    else huffCodes.clear();
  }
  else if (hd.tryHuffmanFlt())    // Lerc2.cpp:305-328
  {
    huffCodes.clear();
    if (!fplCompute(data, hd.nCols, hd.nRows, hd.nDepth, fpl)) return 0;
    nBytesHuff = fpl.compressedLength();
    if (nBytesHuff < 0) nBytesHuff = INT_MAX;
    if (nBytesHuff < nBytesTiling * 0.9) { nBytesData = nBytesHuff; imageMode = IEM_DELTADELTA_HUFFMAN; }    // at least 10 % better
  }

  oneSweep = false;
  const size_t nBytesOneSweep = sizeof(T) * nD * (size_t)numValid;

  // retry with 16x16 blocks when the bit rate is low (Lerc2.cpp:333-357)
  if (((size_t)nBytesTiling * 8 < (size_t)numTotal * nD * 1.5)
    && ((size_t)nBytesTiling < 4 * nBytesOneSweep)
    && (nBytesHuff == 0 || (size_t)nBytesTiling < (size_t)2 * nBytesHuff)
    && (hd.nRows > 8 || hd.nCols > 8))
  {
    hd.mbSize = 16;
    int64_t n2 = 0;
    if (!tilesPass<T>(data, nullptr, n2) || n2 > INT_MAX) return 0;
    if ((int)n2 <= nBytesData) { nBytesData = (int)n2; imageMode = IEM_TILING; huffCodes.clear(); }
    else hd.mbSize = 8;
  }
  if (hd.tryHuffmanInt() || hd.tryHuffmanFlt()) nBytesData += 1;

  size_t total = (size_t)hd.blobSize;
  if (nBytesOneSweep <= (size_t)nBytesData) { oneSweep = true; total += 1 + nBytesOneSweep; }
  else { oneSweep = false; total += 1 + (size_t)nBytesData; }
  if (total > (size_t)INT_MAX) return 0;
  hd.blobSize = (int)total;
  return (unsigned)hd.blobSize;
}

// Lerc2.cpp:396-480 (+ :921-957 mask, :2610-2638 ranges, :1343-1364 one sweep, :1012-1030 checksum)
template<class T> bool Band::emit(const T* data, Writer& w)
{
  u8* blob = w.p;
  writeHeader(w, hd);
  {
    const int numTotal = hd.nCols * hd.nRows;
    const bool needMask = hd.numValid > 0 && hd.numValid < numTotal;
    if (needMask && encodeMask)
    {
      std::vector<u8> rle;
      rleEncode(mask.bits.data(), mask.nBytes(), rle);
      int n = (int)rle.size();
      w.put(&n, 4);
      w.put(rle.data(), rle.size());
    }
    else { int z = 0; w.put(&z, 4); }
  }
  auto finish = [&]() -> bool
  {
    if ((size_t)(w.p - blob) != (size_t)hd.blobSize) return false;
    if (hd.version >= 3)
    {
      unsigned cs = fletcher32(blob + 14, hd.blobSize - 14);
      memcpy(blob + 10, &cs, 4);
    }
    return true;
  };
  if (hd.numValid == 0 || hd.zMin == hd.zMax) return finish();

  const int nD = hd.nDepth;
  if (hd.version >= 4)
  {
    for (int m = 0; m < nD; m++) { T v = (T)zMinVec[m]; w.put(&v, sizeof(T)); }
    for (int m = 0; m < nD; m++) { T v = (T)zMaxVec[m]; w.put(&v, sizeof(T)); }
    if (0 == memcmp(zMinVec.data(), zMaxVec.data(), nD * sizeof(double))) return finish();
  }
  w.byte(oneSweep ? 1 : 0);
  if (oneSweep)
  {
    for (int64_t k = 0, n = (int64_t)hd.nRows * hd.nCols; k < n; k++)
      if (mask.valid(k)) w.put(data + k * nD, nD * sizeof(T));
    return finish();
  }
  if (hd.tryHuffmanInt() || hd.tryHuffmanFlt())
  {
    w.byte((u8)imageMode);
    if (imageMode != IEM_TILING)
    {
      if (hd.tryHuffmanFlt())
      {
        if (imageMode != IEM_DELTADELTA_HUFFMAN) return false;
        fplWrite(fpl, w);
        return finish();
      }
      if (huffCodes.empty()) return false;
      if (!huffEncode(data, w)) return false;
      return finish();
    }
  }
  int64_t n = 0;
  if (!tilesPass<T>(data, &w, n)) return false;
  return finish();
}

// Lerc2.cpp:961-1008
bool Band::readMask(Reader& r)
{
  const int nv = hd.numValid, w = hd.nCols, h = hd.nRows;
  int nm;
  if (!r.get(&nm, 4) || nm < 0) return false;
  if ((nv == 0 || nv == w * h) && nm != 0) return false;
  bool fresh = (mask.nCols != w || mask.nRows != h);
  mask.resize(w, h);
  if (nv == 0) mask.fill(false);
  else if (nv == w * h) mask.fill(true);
  else if (nm > 0)
  {
    if (r.left < (size_t)nm) return false;
    if (!rleDecode(r.p, r.left, mask.bits.data(), mask.nBytes())) return false;
    r.skip(nm);
  }
  else if (fresh) return false;    // "use previous mask" without a previous one (reference: uninitialised bits)
  return true;
}

// Lerc2.cpp:2681-2721
template<class T> bool Band::fillConst(T* data) const
{
  const int nD = hd.nDepth;
  std::vector<T> px(nD, (T)hd.zMin);
  if (nD > 1 && hd.zMin != hd.zMax)
  {
    if ((int)zMinVec.size() != nD) return false;
    for (int m = 0; m < nD; m++) px[m] = (T)zMinVec[m];
  }
  for (int64_t k = 0, n = (int64_t)hd.nRows * hd.nCols; k < n; k++)
    if (mask.valid(k)) memcpy(data + k * nD, px.data(), nD * sizeof(T));
  return true;
}

// Lerc2.cpp:2025-2230
template<class T> bool Band::readBlock(Reader& r0, T* data, int i0, int i1, int j0, int j1, int iDepth,
  std::vector<unsigned>& buf) const
{
  Reader r = r0;
  u8 flag;
  if (!r.get(&flag, 1)) return false;
  const int nCols = hd.nCols, nD = hd.nDepth;
  const bool diff = (hd.version >= 5) ? (flag & 4) != 0 : false;
  const int pattern = (hd.version >= 5) ? 14 : 15;
  if (((flag >> 2) & pattern) != ((j0 >> 3) & pattern)) return false;
  if (diff && iDepth == 0) return false;
  const int tc = flag >> 6, mode = flag & 3;

  auto forEachValid = [&](auto&& fn)
  {
    for (int i = i0; i < i1; i++)
      for (int j = j0; j < j1; j++)
      {
        int64_t k = (int64_t)i * nCols + j;
        if (mask.valid(k)) fn(k * nD + iDepth);
      }
  };

  if (mode == 2)
    forEachValid([&](int64_t m) { data[m] = diff ? data[m - 1] : 0; });
  else if (mode == 0)
  {
    if (diff) return false;
    bool ok = true;
    forEachValid([&](int64_t m)
    {
      if (!ok) return;
      T v;
      if (!r.get(&v, sizeof(T))) { ok = false; return; }
      data[m] = v;
    });
    if (!ok) return false;
  }
  else
  {
    int dtU = typeUsed((diff && hd.dt < DT_FLOAT) ? DT_INT : hd.dt, tc);
    if (dtU == DT_UNDEF) return false;
    size_t n = dtSize(dtU);
    if (r.left < n) return false;
    double offset = getTyped(r.p, dtU);
    r.skip(n);
    double zMax = (hd.version >= 4 && nD > 1) ? zMaxVec[iDepth] : hd.zMax;
    if (mode == 3)
    {
      forEachValid([&](int64_t m)
      {
        if (!diff) data[m] = (T)offset;
        else { double z = offset + data[m - 1]; data[m] = (T)std::min(z, zMax); }
      });
    }
    else
    {
      size_t cap = (size_t)(i1 - i0) * (j1 - j0);
      if (!decodeBitStuffer(r, buf, cap, hd.version)) return false;
      double inv = 2 * hd.maxZErr;
      size_t at = 0;
      bool ok = true;
      forEachValid([&](int64_t m)
      {
        if (at >= buf.size()) { ok = false; return; }    // reference reads past the buffer here
        double z = offset + buf[at++] * inv + (diff ? data[m - 1] : 0);
        data[m] = (T)std::min(z, zMax);
      });
      if (!ok) return false;
    }
  }
  r0 = r;
  return true;
}

// Lerc2.cpp:1672-1713
template<class T> bool Band::readTiles(Reader& r, T* data) const
{
  const int mb = hd.mbSize, nD = hd.nDepth;
  if (mb > 32) return false;
  std::vector<unsigned> buf;
  const int nTV = (hd.nRows + mb - 1) / mb, nTH = (hd.nCols + mb - 1) / mb;
  for (int it = 0; it < nTV; it++)
    for (int jt = 0; jt < nTH; jt++)
      for (int iD = 0; iD < nD; iD++)
      {
        int i0 = it * mb, j0 = jt * mb;
        if (!readBlock(r, data, i0, std::min(hd.nRows, i0 + mb), j0, std::min(hd.nCols, j0 + mb), iD, buf)) return false;
      }
  return true;
}

// Lerc2.cpp:577-694
template<class T> bool Band::decode(Reader& r, T* data, u8* maskBitsOut)
{
  if (!data) return false;
  const u8* blob = r.p;
  const size_t left0 = r.left;
  if (!readHeader(r, hd)) return false;
  if (left0 < (size_t)hd.blobSize) return false;
  if (hd.dt != DtOf<T>::v) return false;    // reference leaves this to the caller; we refuse
  if (hd.version >= 3)
  {
    if (hd.blobSize < 14) return false;
    if (fletcher32(blob + 14, hd.blobSize - 14) != hd.checksum) return false;
  }
  if (!readMask(r)) return false;
  if (maskBitsOut) memcpy(maskBitsOut, mask.bits.data(), mask.nBytes());
  memset(data, 0, (size_t)hd.nCols * hd.nRows * hd.nDepth * sizeof(T));
  if (hd.numValid == 0) return true;
  if (hd.zMin == hd.zMax) return fillConst(data);

  const int nD = hd.nDepth;
  if (hd.version >= 4)
  {
    zMinVec.resize(nD); zMaxVec.resize(nD);
    std::vector<T> tmp(nD);
    if (!r.get(tmp.data(), nD * sizeof(T))) return false;
    for (int m = 0; m < nD; m++) zMinVec[m] = tmp[m];
    if (!r.get(tmp.data(), nD * sizeof(T))) return false;
    for (int m = 0; m < nD; m++) zMaxVec[m] = tmp[m];
    if (0 == memcmp(zMinVec.data(), zMaxVec.data(), nD * sizeof(double))) return fillConst(data);
  }
  u8 sweep;
  if (!r.get(&sweep, 1)) return false;
  if (sweep)
  {
    int64_t nv = mask.countValid();
    if (nv < 0 || r.left < (size_t)nv * nD * sizeof(T)) return false;
    for (int64_t k = 0, n = (int64_t)hd.nRows * hd.nCols; k < n; k++)
      if (mask.valid(k)) r.get(data + k * nD, nD * sizeof(T));
    return true;
  }
  if (hd.tryHuffmanInt() || hd.tryHuffmanFlt())
  {
    u8 f;
    if (!r.get(&f, 1)) return false;
    if (f > 3 || (f > 2 && hd.version < 6) || (f > 1 && hd.version < 4)) return false;
    imageMode = f;
    if (imageMode != IEM_TILING)
    {
      if (hd.tryHuffmanInt())
      {
        if (imageMode == IEM_DELTA_HUFFMAN || (hd.version >= 4 && imageMode == IEM_HUFFMAN)) return huffDecode(r, data);
        return false;
      }
      if (hd.tryHuffmanFlt() && imageMode == IEM_DELTADELTA_HUFFMAN)    // Lerc2.cpp:674-678
        return fplDecode(r, data, hd.nCols, hd.nRows, hd.nDepth);
      return false;
    }
  }
  return readTiles(r, data);
}

// Lerc2.cpp:516-573
bool Band::ranges(const u8* p, size_t n, double* mins, double* maxs)
{
  Reader r{ p, n };
  if (!readHeader(r, hd) || hd.version < 4) return false;
  if (!readMask(r)) return false;
  const int nD = hd.nDepth;
  if (hd.numValid == 0) { for (int m = 0; m < nD; m++) mins[m] = maxs[m] = 0; return true; }
  if (hd.zMin == hd.zMax) { for (int m = 0; m < nD; m++) mins[m] = maxs[m] = hd.zMin; return true; }
  const size_t sz = dtSize(hd.dt);
  if (r.left < 2 * sz * nD) return false;
  for (int m = 0; m < nD; m++) mins[m] = getTyped(r.p + m * sz, hd.dt);
  for (int m = 0; m < nD; m++) maxs[m] = getTyped(r.p + (nD + m) * sz, hd.dt);
  return true;
}

// ---------------------------------------------------------------------------------------------
// facade -- Lerc.cpp: CheckDimensions :1622-1639, FilterNoDataAndNaN :1378-1552 (noData-free
// subset + noData), EncodeInternal :628-789, GetLercInfo :92-182, DecodeTempl :397-521
// ---------------------------------------------------------------------------------------------
static bool dimsOk(int nDepth, int nCols, int nRows, size_t elemSize)
{
  if (nDepth <= 0 || nCols <= 0 || nRows <= 0) return false;
  const uint64_t nPix = (uint64_t)nRows * nCols, lim = INT_MAX, bpp = elemSize;
  return !(nPix > lim || bpp > lim || bpp * nDepth > lim || bpp * nDepth * nPix > lim);
}

template<class T> static bool isIntVal(T z) { return z == (T)floor((double)z + 0.5); }    // Lerc.h:271

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
      if ((v > (T)lowIntLimit) && (v < (T)(minVal - 2 * maxZErr)) && isIntVal(v)) { out = v; return true; }
  }
  else
  {
    const double dist[] = { 4 * maxZErr, 0.0001, 0.001, 0.01, 0.1, 1, 10, 100, 1000, 10000 };
    for (double d : dist) cand.push_back((T)(minVal - d));
    cand.push_back((T)(minVal > 0 ? minVal / 2 : minVal * 2));
    std::sort(cand.begin(), cand.end(), std::greater<double>());
    T lowest = (T)(std::is_same<T, float>::value ? -FLT_MAX : -DBL_MAX);
    for (T v : cand)
      if ((v > lowest) && (v < (T)(minVal - 2 * maxZErr))) { out = v; return true; }
  }
  return false;
}

// float / double only
template<class T>
static Err filterNoDataAndNaN(std::vector<T>& data, std::vector<u8>& mask, int nDepth, int nCols, int nRows,
  double& maxZErr, bool passNoData, double& noDataValue, bool& modifiedMask, bool& needNoData, bool& allIntOut,
  double& minOut, double& maxOut)
{
  modifiedMask = needNoData = allIntOut = false;
  const bool isF32 = std::is_same<T, float>::value;
  bool noDataLeft = false, allInt = true, hasNaN = false;
  T origNoData(0);
  if (passNoData)
  {
    if (isF32 && (noDataValue < -FLT_MAX || noDataValue > FLT_MAX)) return WRONG_PARAM;
    origNoData = (T)noDataValue;
  }
  else origNoData = (T)(isF32 ? -FLT_MAX : -DBL_MAX);
  const double lowInt = isF32 ? -(double)(1L << 23) : -(double)((int64_t)1 << 53);
  const double highInt = -lowInt;
  double minVal = DBL_MAX, maxVal = -DBL_MAX;

  for (int64_t k = 0, n = (int64_t)nRows * nCols; k < n; k++)
  {
    if (!mask[k]) continue;
    int bad = 0;
    for (int m = 0; m < nDepth; m++)
    {
      T& z = data[k * nDepth + m];
      if (std::isnan((double)z))
      {
        hasNaN = true; bad++;
        if (passNoData && nDepth > 1) z = origNoData; else if (nDepth == 1) z = 0;
      }
      else if (passNoData && z == origNoData) bad++;
      else
      {
        if (z < minVal) minVal = z;
        if (z > maxVal) maxVal = z;
        if (allInt && !isIntVal(z)) allInt = false;
      }
    }
    if (bad == nDepth) { mask[k] = 0; modifiedMask = true; }
    else if (bad > 0) noDataLeft = true;
  }
  if (minVal == DBL_MAX && maxVal == -DBL_MAX) { minOut = maxOut = 0; maxZErr = 0; return OK; }
  minOut = minVal; maxOut = maxVal;
  needNoData = noDataLeft;
  if (hasNaN && nDepth > 1 && noDataLeft && !passNoData) return ERR_NAN;

  double e = maxZErr;
  if (allInt)
  {
    allInt &= (minVal >= lowInt) && (minVal <= highInt) && (maxVal >= lowInt) && (maxVal <= highInt);
    if (noDataLeft) allInt &= isIntVal(origNoData) && (origNoData >= lowInt) && (origNoData <= highInt);
    if (allInt) e = std::max(0.5, floor(maxZErr));
  }
  allIntOut = allInt;
  if (e == 0) return OK;
  if (passNoData)
  {
    double dist = allInt ? floor(e) : 2 * e;
    if ((origNoData >= minVal - dist) && (origNoData <= maxVal + dist)) { maxZErr = allInt ? 0.5 : 0; return OK; }
  }
  if (noDataLeft)
  {
    T remap = origNoData;
    if (findNoDataBelowMin(minVal, e, allInt, lowInt, remap))
    {
      if (remap != origNoData)
      {
        for (int64_t k = 0, n = (int64_t)nRows * nCols; k < n; k++)
          if (mask[k])
            for (int m = 0; m < nDepth; m++)
              if (data[k * nDepth + m] == origNoData) data[k * nDepth + m] = remap;
        noDataValue = remap;
      }
    }
    else if ((double)origNoData >= minVal) e = allInt ? 0.5 : 0;
  }
  if (maxZErr != e) maxZErr = e;
  return OK;
}

template<class T> static bool typeRange(std::pair<double, double>& r)
{
  if (std::is_same<T, unsigned char>::value) r = { 0, UCHAR_MAX };
  else if (std::is_same<T, unsigned short>::value) r = { 0, USHRT_MAX };
  else if (std::is_same<T, unsigned int>::value) r = { 0, UINT_MAX };
  else if (std::is_same<T, signed char>::value) r = { CHAR_MIN, CHAR_MAX };
  else if (std::is_same<T, short>::value) r = { SHRT_MIN, SHRT_MAX };
  else if (std::is_same<T, int>::value) r = { INT_MIN, INT_MAX };
  else return false;
  return true;
}

// integer types with a noData value -- Lerc.cpp:1241-1374
template<class T>
static Err filterNoDataInt(std::vector<T>& data, std::vector<u8>& mask, int nDepth, int nCols, int nRows,
  double& maxZErr, bool passNoData, double& noDataValue, bool& modifiedMask, bool& needNoData, double& minOut, double& maxOut)
{
  modifiedMask = needNoData = false;
  if (!passNoData) return OK;
  std::pair<double, double> tr;
  if (!typeRange<T>(tr)) return FAILED;
  if (noDataValue < tr.first || noDataValue > tr.second) return WRONG_PARAM;
  T orig = (T)noDataValue;
  double minVal = DBL_MAX, maxVal = -DBL_MAX;
  for (int64_t k = 0, n = (int64_t)nRows * nCols; k < n; k++)
  {
    if (!mask[k]) continue;
    int bad = 0;
    for (int m = 0; m < nDepth; m++)
    {
      T z = data[k * nDepth + m];
      if (z == orig) bad++;
      else { if (z < minVal) minVal = z; if (z > maxVal) maxVal = z; }
    }
    if (bad == nDepth) { mask[k] = 0; modifiedMask = true; }
    else if (bad > 0) needNoData = true;
  }
  double e = std::max(0.5, floor(maxZErr));
  double dist = floor(e);
  if (minVal == DBL_MAX && maxVal == -DBL_MAX) { minOut = maxOut = 0; maxZErr = 0.5; return OK; }
  minOut = minVal; maxOut = maxVal;
  if ((orig >= minVal - dist) && (orig <= maxVal + dist)) { maxZErr = 0.5; return OK; }
  if (needNoData)
  {
    double minDist = floor(e) + 1;
    double remap = minVal - minDist;
    T nd = orig;
    if (remap >= tr.first) nd = (T)remap;
    else
    {
      e = 0.5;
      remap = minVal - 1;
      if (remap >= tr.first) nd = (T)remap;
      else
      {
        remap = maxVal + 1;
        if ((remap <= tr.second) && (remap < orig)) nd = (T)remap;
      }
    }
    if (nd != orig)
    {
      for (int64_t k = 0, n = (int64_t)nRows * nCols; k < n; k++)
        if (mask[k])
          for (int m = 0; m < nDepth; m++)
            if (data[k * nDepth + m] == orig) data[k * nDepth + m] = nd;
      noDataValue = nd;
    }
  }
  if (maxZErr != e) maxZErr = e;
  return OK;
}

static void bytesToBits(const u8* byteMask, int nCols, int nRows, Mask& m)    // Lerc.cpp:959-975
{
  m.resize(nCols, nRows);
  m.fill(true);
  for (int64_t k = 0, n = (int64_t)nCols * nRows; k < n; k++) if (!byteMask[k]) m.clear(k);
}

template<class T>
static Err encodeBands(const T* pData, int version, int nDepth, int nCols, int nRows, int nBands, int nMasks,
  const u8* pValidBytes, double maxZErr, unsigned& numBytesNeeded, u8* pBuffer, unsigned numBytesBuffer,
  unsigned& numBytesWritten, const u8* pUsesNoData, const double* noDataValues)
{
  numBytesNeeded = numBytesWritten = 0;
  if (version >= 0 && version != kCurrentVersion) return WRONG_PARAM;    // older codecs: encodeBandsOld() below
  if (pUsesNoData && !noDataValues)
    for (int i = 0; i < nBands; i++) if (pUsesNoData[i]) return WRONG_PARAM;

  Band band;
  u8* dst = pBuffer;
  const size_t nPix = (size_t)nCols * nRows, nElem = nPix * nDepth;
  std::vector<T> data(nElem);
  std::vector<u8> mask(nPix), prevMask;
  bool havePrev = false, anyMaskModified = false;
  Mask bitMask;
  const bool isFlt = std::is_floating_point<T>::value;

  for (int iBand = 0; iBand < nBands; iBand++)
  {
    bool encMask = (iBand == 0);
    const T* arr = pData + nElem * iBand;
    const u8* bm = (nMasks > 0) ? (pValidBytes + ((nMasks > 1) ? nPix * iBand : 0)) : nullptr;
    memcpy(data.data(), arr, nElem * sizeof(T));
    if (bm) memcpy(mask.data(), bm, nPix); else memset(mask.data(), 1, nPix);

    double e = maxZErr;
    const bool passNoData = pUsesNoData && pUsesNoData[iBand] > 0;
    const double noDataOrig = passNoData ? noDataValues[iBand] : 0;
    double noDataL = noDataOrig;
    bool allInt = false, modMask = false, needNoData = false;
    double minVal = +1, maxVal = -1;
    band.zMinVec.clear(); band.zMaxVec.clear(); band.minMaxSet = false;
    Err rc = OK;
    if (isFlt)
      rc = filterNoDataAndNaN(data, mask, nDepth, nCols, nRows, e, passNoData, noDataL, modMask, needNoData, allInt, minVal, maxVal);
    else if (passNoData)
      rc = filterNoDataInt(data, mask, nDepth, nCols, nRows, e, passNoData, noDataL, modMask, needNoData, minVal, maxVal);
    if (rc != OK) return rc;
    if (modMask) anyMaskModified = true;
    const bool compareMasks = (nMasks > 1) || anyMaskModified;
    if (compareMasks && iBand > 0 && havePrev && memcmp(mask.data(), prevMask.data(), nPix)) encMask = true;
    if (nBands > 1 && iBand < nBands - 1) { prevMask = mask; havePrev = true; }

    if (encMask)
    {
      bool allValid = !memchr(mask.data(), 0, nPix);
      if (!allValid) bytesToBits(mask.data(), nCols, nRows, bitMask);
      if (!band.setDims(nDepth, nCols, nRows, allValid ? nullptr : bitMask.bits.data())) return FAILED;
    }
    band.hd.passNoData = needNoData;
    band.hd.noDataVal = needNoData ? noDataL : 0;
    band.hd.noDataValOrig = needNoData ? noDataOrig : 0;
    band.hd.nBlobsMore = nBands - 1 - iBand;
    band.hd.isInt = allInt ? 1 : 0;
    if (nDepth == 1 && maxVal >= minVal)
    {
      band.zMinVec.assign(1, minVal); band.zMaxVec.assign(1, maxVal); band.minMaxSet = true;
    }
    unsigned nBytes = band.plan(data.data(), e, encMask);
    if (nBytes == 0) return FAILED;
    if ((size_t)numBytesNeeded + nBytes > (size_t)UINT_MAX) return DIMS_TOO_LARGE;
    numBytesNeeded += nBytes;
    if (pBuffer)
    {
      if ((size_t)(dst - pBuffer) + nBytes > numBytesBuffer) return BUFFER_TOO_SMALL;
      Writer w{ dst };
      if (!band.emit(data.data(), w)) return FAILED;
      dst = w.p;
    }
  }
  numBytesWritten = (unsigned)(dst - pBuffer);
  return OK;
}

// Lerc::EncodeInternal_v5 (Lerc.cpp:526-624): what lerc_encodeForVersion does for codec versions 3..5.  No noData
// filter and no all-integer promotion; a NaN becomes -FLT_MAX / -DBL_MAX, and a pixel that is NaN in every depth
// leaves the mask (CheckForNaN :861-897, ReplaceNaNValues :901-938).  Lossless float has no Huffman mode before
// codec 6 (Lerc2.h:130): it goes through the tiling with raw blocks.  Codec 2 has no checksum and packs bits the old
// way (stuffBitsOld).
template<class T>
static Err encodeBandsOld(const T* pData, int version, int nDepth, int nCols, int nRows, int nBands, int nMasks,
  const u8* pValidBytes, double maxZErr, unsigned& numBytesNeeded, u8* pBuffer, unsigned numBytesBuffer, unsigned& numBytesWritten)
{
  numBytesNeeded = numBytesWritten = 0;
  if (version < 2 || version > 5) return WRONG_PARAM;              // Lerc2.cpp:52-62
  if (version < 4 && nDepth > 1) return FAILED;                   // Lerc2::Set refuses (Lerc2.cpp:85-86)
  Band band;
  band.hd.version = version;
  u8* dst = pBuffer;
  const size_t nPix = (size_t)nCols * nRows, nElem = nPix * nDepth;
  std::vector<T> data(nElem);
  std::vector<u8> mask(nPix), prevMask;
  bool havePrev = false;
  Mask bitMask;
  const bool isFlt = std::is_floating_point<T>::value;
  const T nanStandIn = (T)(std::is_same<T, float>::value ? -FLT_MAX : -DBL_MAX);

  for (int iBand = 0; iBand < nBands; iBand++)
  {
    bool encMask = (iBand == 0);
    const T* arr = pData + nElem * iBand;
    const u8* bm = (nMasks > 0) ? (pValidBytes + ((nMasks > 1) ? nPix * iBand : 0)) : nullptr;
    memcpy(data.data(), arr, nElem * sizeof(T));
    if (bm) memcpy(mask.data(), bm, nPix); else memset(mask.data(), 1, nPix);
    if (isFlt)
      for (size_t k = 0; k < nPix; k++)
      {
        if (!mask[k]) continue;
        int cntNaN = 0;
        for (int m = 0; m < nDepth; m++)
          if (std::isnan((double)data[k * nDepth + m])) { cntNaN++; data[k * nDepth + m] = nanStandIn; }
        if (cntNaN == nDepth) mask[k] = 0;
      }
    if (iBand > 0 && havePrev && memcmp(mask.data(), prevMask.data(), nPix)) encMask = true;    // MasksDiffer, Lerc.cpp:572,586
    if (iBand < nBands - 1) { prevMask = mask; havePrev = true; }
    if (encMask)
    {
      const bool allValid = !memchr(mask.data(), 0, nPix);
      if (!allValid) bytesToBits(mask.data(), nCols, nRows, bitMask);
      if (!band.setDims(nDepth, nCols, nRows, allValid ? nullptr : bitMask.bits.data())) return FAILED;
    }
    band.zMinVec.clear(); band.zMaxVec.clear(); band.minMaxSet = false;
    unsigned nBytes = band.plan(data.data(), maxZErr, encMask);
    if (nBytes == 0) return FAILED;
    if ((size_t)numBytesNeeded + nBytes > (size_t)UINT_MAX) return DIMS_TOO_LARGE;
    numBytesNeeded += nBytes;
    if (pBuffer)
    {
      if ((size_t)(dst - pBuffer) + nBytes > numBytesBuffer) return BUFFER_TOO_SMALL;
      Writer w{ dst };
      if (!band.emit(data.data(), w)) return FAILED;
      dst = w.p;
    }
  }
  numBytesWritten = (unsigned)(dst - pBuffer);
  return OK;
}

struct Info
{
  int version = 0, nDepth = 0, nCols = 0, nRows = 0, numValid = 0, nBands = 0, nMasks = 0, nUsesNoData = 0, dt = 0;
  unsigned blobSize = 0;
  double zMin = 0, zMax = 0, maxZErr = 0;
};

static Err bandRanges(const u8* p, unsigned n, int iBand, const Header& h, double* mins, double* maxs, size_t nElem)
{
  const int nD = h.nDepth;
  if (nD <= 0 || iBand < 0 || !mins || !maxs) return WRONG_PARAM;
  if (nElem < ((size_t)iBand + 1) * (size_t)nD) return BUFFER_TOO_SMALL;
  if (nD == 1) { mins[iBand] = h.zMin; maxs[iBand] = h.zMax; return OK; }
  if (h.passNoData) return HAS_NODATA;
  Band b;
  return b.ranges(p, n, mins + (size_t)iBand * nD, maxs + (size_t)iBand * nD) ? OK : FAILED;
}

// ---------------------------------------------------------------------------------------------
// Lerc1 ("CntZImage", decode only) -- Lerc1Decode/CntZImage.cpp:74-480, Lerc1Decode/BitStuffer.cpp:32-157.
// Blob = "CntZImage " | version 11 | type 8 | height | width | maxZError | count part | z part; every further band is
// header + z part only.  A part = numTilesVert | numTilesHori | numBytes | maxValInImg | tiles.  Counts are, in practice, a
// validity mask (RLE of the bit mask) or constant; z tiles are raw floats, a constant, or offset + bit-stuffed integers
// (the MSB-first word layout codec 2 inherited).  Unlike the reference, every read is bounds checked.
// ---------------------------------------------------------------------------------------------
struct Lerc1Image
{
  int width = 0, height = 0;
  double maxZErr = 0;
  std::vector<float> cnt, z;
  bool ignoreMask = false;    // m_bDecoderCanIgnoreMask
};

static bool lerc1ReadFlt(Reader& r, float& z, int numBytes)    // CntZImage.cpp:449-477
{
  if (numBytes == 1) { signed char c; if (!r.get(&c, 1)) return false; z = c; return true; }
  if (numBytes == 2) { short v; if (!r.get(&v, 2)) return false; z = v; return true; }
  if (numBytes == 4) return r.get(&z, 4);
  return false;
}

static bool lerc1BitStuffer(Reader& r, std::vector<unsigned>& data)    // BitStuffer.cpp:32-112
{
  u8 b0;
  if (!r.get(&b0, 1)) return false;
  const int bits67 = b0 >> 6, nb = b0 & 63;
  const int nCount = (bits67 == 0) ? 4 : 3 - bits67;
  unsigned n = 0;
  if (nCount == 1) { u8 c; if (!r.get(&c, 1)) return false; n = c; }
  else if (nCount == 2) { unsigned short v; if (!r.get(&v, 2)) return false; n = v; }
  else if (nCount == 4) { if (!r.get(&n, 4)) return false; }
  else return false;
  if (nb >= 32) return false;
  data.assign(n, 0u);
  if (n == 0 || nb == 0) return true;
  std::vector<unsigned> tmp;
  if (!unstuffBitsOld(r, tmp, n, nb)) return false;
  data.swap(tmp);
  return true;
}

static bool lerc1ReadCntTile(Reader& r, Lerc1Image& img, int i0, int i1, int j0, int j1)    // CntZImage.cpp:262-337
{
  u8 flag;
  if (i0 >= i1 || j0 >= j1 || !r.get(&flag, 1)) return false;
  if (flag == 2) return true;    // all 0 (the image was cleared)
  if (flag == 3 || flag == 4)
  {
    for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) { img.cnt[(size_t)i * img.width + j] = (flag == 3) ? -1.0f : 1.0f; img.z[(size_t)i * img.width + j] = 0; }
    return true;
  }
  if ((flag & 63) > 4) return false;
  if (flag == 0)
  {
    for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) if (!r.get(&img.cnt[(size_t)i * img.width + j], 4)) return false;
    return true;
  }
  const int bits67 = flag >> 6, n = (bits67 == 0) ? 4 : 3 - bits67;
  float offset = 0;
  std::vector<unsigned> data;
  if (!lerc1ReadFlt(r, offset, n) || !lerc1BitStuffer(r, data) || data.size() < (size_t)(i1 - i0) * (j1 - j0)) return false;
  size_t k = 0;
  for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) img.cnt[(size_t)i * img.width + j] = offset + (float)data[k++];
  return true;
}

static bool lerc1ReadZTile(Reader& r, Lerc1Image& img, int i0, int i1, int j0, int j1, float maxZInImg)    // CntZImage.cpp:341-438
{
  u8 flag;
  if (!r.get(&flag, 1)) return false;
  const int bits67 = flag >> 6;
  flag &= 63;
  auto at = [&](int i, int j) { return (size_t)i * img.width + j; };
  if (flag == 2)
  {
    for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) if (img.cnt[at(i, j)] > 0) img.z[at(i, j)] = 0;
    return true;
  }
  if (flag > 3) return false;
  if (flag == 0)
  {
    for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) if (img.cnt[at(i, j)] > 0 && !r.get(&img.z[at(i, j)], 4)) return false;
    return true;
  }
  const int n = (bits67 == 0) ? 4 : 3 - bits67;
  float offset = 0;
  if (!lerc1ReadFlt(r, offset, n)) return false;
  if (flag == 3)
  {
    for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) if (img.cnt[at(i, j)] > 0) img.z[at(i, j)] = offset;
    return true;
  }
  std::vector<unsigned> data;
  if (!lerc1BitStuffer(r, data)) return false;
  const double invScale = 2 * img.maxZErr;
  size_t k = 0;
  for (int i = i0; i < i1; i++)
    for (int j = j0; j < j1; j++)
      if (img.ignoreMask || img.cnt[at(i, j)] > 0)
      {
        if (k >= data.size()) return false;
        const float z = (float)(offset + data[k++] * invScale);
        img.z[at(i, j)] = std::min(z, maxZInImg);
      }
  return true;
}

static const size_t kLerc1HeaderBytes = 10 + 4 * 4 + 8;

// CntZImage::read (CntZImage.cpp:74-215)
static bool lerc1Read(Reader& r, Lerc1Image& img, bool onlyHeader, bool onlyZPart)
{
  if (r.left < kLerc1HeaderBytes || memcmp(r.p, "CntZImage ", 10)) return false;
  r.skip(10);
  int version = 0, type = 0, width = 0, height = 0;
  double maxZErr = 0;
  r.get(&version, 4); r.get(&type, 4); r.get(&height, 4); r.get(&width, 4); r.get(&maxZErr, 8);
  if (version != 11 || type != 8) return false;
  if (height < 0 || width < 0 || height > 40000 || width > 40000) return false;
  if ((size_t)8 * height * width > (size_t)INT_MAX) return false;
  if (maxZErr > 1e12) return false;
  if (onlyHeader) { img.width = width; img.height = height; img.maxZErr = maxZErr; return true; }
  if (!onlyZPart) { img.width = width; img.height = height; img.cnt.assign((size_t)width * height, 0.f); img.z.assign((size_t)width * height, 0.f); }
  else if (width != img.width || height != img.height) return false;
  img.maxZErr = maxZErr;
  if (!onlyZPart) img.ignoreMask = false;
  for (int iPart = onlyZPart ? 1 : 0; iPart < 2; iPart++)
  {
    const bool zPart = iPart == 1;
    int nTV = 0, nTH = 0, numBytes = 0;
    float maxVal = 0;
    if (r.left < 16) return false;
    r.get(&nTV, 4); r.get(&nTH, 4); r.get(&numBytes, 4); r.get(&maxVal, 4);
    if (numBytes < 0 || (size_t)numBytes > r.left) return false;
    Reader part{ r.p, (size_t)numBytes };
    if (!zPart && nTV == 0 && nTH == 0)
    {
      if (numBytes == 0)
      {
        std::fill(img.cnt.begin(), img.cnt.end(), maxVal);
        if (maxVal > 0) img.ignoreMask = true;
      }
      else
      {
        Mask m;
        m.resize(width, height);
        if (!rleDecode(part.p, part.left, m.bits.data(), m.nBytes())) return false;
        for (int64_t k = 0, nPix = (int64_t)width * height; k < nPix; k++) img.cnt[k] = m.valid(k) ? 1.0f : 0.0f;
      }
    }
    else
    {
      if (nTV <= 0 || nTH <= 0 || nTV > height || nTH > width) return false;    // readTiles, CntZImage.cpp:219-258
      for (int it = 0; it <= nTV; it++)
      {
        int tileH = height / nTV;
        const int i0 = it * tileH;
        if (it == nTV) tileH = height % nTV;
        if (tileH == 0) continue;
        for (int jt = 0; jt <= nTH; jt++)
        {
          int tileW = width / nTH;
          const int j0 = jt * tileW;
          if (jt == nTH) tileW = width % nTH;
          if (tileW == 0) continue;
          const bool ok = zPart ? lerc1ReadZTile(part, img, i0, i0 + tileH, j0, j0 + tileW, maxVal)
                                : lerc1ReadCntTile(part, img, i0, i0 + tileH, j0, j0 + tileW);
          if (!ok) return false;
        }
      }
    }
    r.skip((size_t)numBytes);
  }
  return true;
}

// Lerc::GetLercInfo, Lerc1 branch (Lerc.cpp:184-266): the bands are decoded to find valid counts and ranges
static Err getInfoLerc1(const u8* blob, unsigned n, Info& info, double* mins, double* maxs)
{
  const size_t hdr0 = kLerc1HeaderBytes + 2 * 16 + 1, hdr1 = kLerc1HeaderBytes + 16 + 1;
  Reader r{ blob, n };
  Lerc1Image img;
  if (hdr0 > n || !lerc1Read(r, img, true, false)) return FAILED;
  info.zMin = FLT_MAX; info.zMax = -FLT_MAX;
  info.nDepth = 1; info.nCols = img.width; info.nRows = img.height; info.dt = DT_FLOAT; info.maxZErr = img.maxZErr;
  r = Reader{ blob, n };
  bool onlyZ = false;
  while ((size_t)info.blobSize + hdr1 < n)
  {
    if (!lerc1Read(r, img, false, onlyZ)) return info.nBands > 0 ? OK : FAILED;
    onlyZ = true;
    info.blobSize = (unsigned)(r.p - blob);
    int numValid = 0;
    float zMin = FLT_MAX, zMax = -FLT_MAX;
    for (size_t k = 0; k < img.cnt.size(); k++)
      if (img.cnt[k] > 0) { numValid++; zMax = std::max(zMax, img.z[k]); zMin = std::min(zMin, img.z[k]); }
    info.numValid = numValid;
    info.zMin = std::min(info.zMin, (double)zMin);
    info.zMax = std::max(info.zMax, (double)zMax);
    info.nMasks = numValid < img.width * img.height ? 1 : 0;
    if (mins && maxs) { mins[info.nBands] = zMin; maxs[info.nBands] = zMax; }
    info.nBands++;
  }
  return OK;
}

static Err getInfo(const u8* blob, unsigned n, Info& info, double* mins = nullptr, double* maxs = nullptr, size_t nElem = 0)
{
  info = Info();
  Header h;
  bool hasMask = false;
  int nMasks = 0;
  if (!peekHeader(blob, n, h, hasMask)) return getInfoLerc1(blob, n, info, mins, maxs);
  if (h.blobSize < 0) return FAILED;
  info.version = h.version; info.nDepth = h.nDepth; info.nCols = h.nCols; info.nRows = h.nRows;
  info.numValid = h.numValid; info.blobSize = (unsigned)h.blobSize; info.dt = h.dt;
  info.zMin = h.zMin; info.zMax = h.zMax; info.maxZErr = h.maxZErr; info.nUsesNoData = h.passNoData ? 1 : 0;
  bool more = (h.version <= 5) || (h.nBlobsMore > 0);
  if (hasMask || info.numValid == 0) nMasks = 1;
  if (mins && maxs) { Err e = bandRanges(blob, n, 0, h, mins, maxs, nElem); if (e != OK) return e; }
  info.nBands = 1;
  if (info.blobSize > n) return FAILED;
  Header hn;
  while (more && peekHeader(blob + info.blobSize, n - info.blobSize, hn, hasMask))
  {
    if (hn.nDepth != info.nDepth || hn.nCols != info.nCols || hn.nRows != info.nRows || hn.dt != info.dt || hn.blobSize < 0)
      return FAILED;
    more = (hn.version <= 5) || (hn.nBlobsMore > 0);
    if (hn.passNoData) info.nUsesNoData++;
    if (hasMask || hn.numValid != info.numValid) nMasks = 2;
    if ((size_t)info.blobSize > (size_t)UINT_MAX - hn.blobSize) return FAILED;
    if ((size_t)info.blobSize + hn.blobSize > (size_t)n) return FAILED;
    info.zMin = std::min(info.zMin, hn.zMin);
    info.zMax = std::max(info.zMax, hn.zMax);
    info.maxZErr = std::max(info.maxZErr, hn.maxZErr);
    if (mins && maxs)
    {
      Err e = bandRanges(blob + info.blobSize, n - info.blobSize, info.nBands, hn, mins, maxs, nElem);
      if (e != OK) return e;
    }
    info.blobSize += hn.blobSize;
    info.nBands++;
  }
  info.nMasks = nMasks > 1 ? info.nBands : nMasks;
  if (info.nUsesNoData > 0) info.nUsesNoData = info.nBands;
  return OK;
}

// Lerc.cpp:1046-1076
template<class T> static void remapNoData(T* data, const Mask& mask, const Header& h)
{
  const T from = (T)h.noDataVal, to = (T)h.noDataValOrig;
  if (from == to) return;
  const bool useMask = (mask.nCols == h.nCols) && (mask.nRows == h.nRows);
  for (int64_t k = 0, n = (int64_t)h.nRows * h.nCols; k < n; k++)
    if (!useMask || mask.valid(k))
      for (int m = 0; m < h.nDepth; m++)
        if (data[k * h.nDepth + m] == from) data[k * h.nDepth + m] = to;
}

template<class T>
static Err decodeBands(T* pData, const u8* blob, unsigned nBytesBlob, int nDepth, int nCols, int nRows, int nBands,
  int nMasks, u8* pValidBytes, u8* pUsesNoData, double* noDataValues)
{
  if (!dimsOk(nDepth, nCols, nRows, sizeof(T))) return DIMS_TOO_LARGE;
  Header h;
  bool hasMask = false;
  if (!peekHeader(blob, nBytesBlob, h, hasMask) || h.version < 1)
  {
    // Lerc1 (Lerc.cpp:487-516 + Lerc::Convert :795-845): pixels that are not valid keep what the caller's buffer held
    const size_t hdr0 = kLerc1HeaderBytes + 2 * 16 + 1, hdr1 = kLerc1HeaderBytes + 16 + 1;
    Reader r{ blob, nBytesBlob };
    Lerc1Image img;
    const bool flt = std::is_floating_point<T>::value;
    for (int iBand = 0; iBand < nBands; iBand++)
    {
      if ((size_t)(r.p - blob) + (iBand == 0 ? hdr0 : hdr1) > nBytesBlob) return FAILED;
      if (!lerc1Read(r, img, false, iBand > 0)) return FAILED;
      if (img.width != nCols || img.height != nRows) return FAILED;
      const size_t nPix = (size_t)nRows * nCols;
      T* arr = pData + (size_t)iBand * nPix;
      u8* msk = iBand < nMasks ? pValidBytes + (size_t)iBand * nPix : nullptr;
      if (msk) memset(msk, 0, nPix);
      for (size_t k = 0; k < nPix; k++)
      {
        if (img.cnt[k] > 0) { arr[k] = flt ? (T)img.z[k] : (T)floor(img.z[k] + 0.5); if (msk) msk[k] = 1; }
        else if (!msk && iBand == 0) return FAILED;
      }
    }
    return OK;
  }
  Info info;
  Err e = getInfo(blob, nBytesBlob, info);
  if (e != OK) return e;
  if (nMasks < info.nMasks) return WRONG_PARAM;
  if (nBands > info.nBands) return WRONG_PARAM;
  if (info.nUsesNoData && nDepth > 1)
  {
    if (!pUsesNoData || !noDataValues) return HAS_NODATA;
    memset(pUsesNoData, 0, nBands);
    memset(noDataValues, 0, nBands * sizeof(double));
  }
  Reader r{ blob, nBytesBlob };
  Band band;
  Mask outMask;
  for (int iBand = 0; iBand < nBands; iBand++)
  {
    if ((size_t)(r.p - blob) >= nBytesBlob || !peekHeader(r.p, r.left, h, hasMask)) continue;
    if (h.nDepth != nDepth || h.nCols != nCols || h.nRows != nRows || h.blobSize < 0) return FAILED;
    if ((size_t)(r.p - blob) + (size_t)h.blobSize > nBytesBlob) return FAILED;
    const size_t nPix = (size_t)iBand * nRows * nCols;
    T* arr = pData + nPix * nDepth;
    const bool getMask = iBand < nMasks;
    if (getMask) outMask.resize(nCols, nRows);
    // the reference advances by what the band decoder consumed; a well formed band consumes blobSize
    Reader rb = r;
    if (!band.decode(rb, arr, getMask ? outMask.bits.data() : nullptr)) return FAILED;
    r = rb;
    if (info.nUsesNoData && nDepth > 1)
    {
      pUsesNoData[iBand] = h.passNoData ? 1 : 0;
      noDataValues[iBand] = h.noDataValOrig;
      if (h.passNoData) remapNoData(arr, band.mask, h);
    }
    if (getMask)
    {
      u8* dstM = pValidBytes + nPix;
      for (int64_t k = 0, n = (int64_t)nCols * nRows; k < n; k++) dstM[k] = outMask.valid(k) ? 1 : 0;
    }
  }
  return OK;
}

}    // namespace orc

// =============================================================================================
// C entry points -- argument checks as Lerc_c_api_impl.cpp:33-304
// =============================================================================================
using namespace orc;

#define ORC_DISPATCH(dt, CALL)                                                     \
  switch (dt) {                                                                    \
    case DT_CHAR:   { typedef signed char    TT; CALL; }                           \
    case DT_BYTE:   { typedef unsigned char  TT; CALL; }                           \
    case DT_SHORT:  { typedef short          TT; CALL; }                           \
    case DT_USHORT: { typedef unsigned short TT; CALL; }                           \
    case DT_INT:    { typedef int            TT; CALL; }                           \
    case DT_UINT:   { typedef unsigned int   TT; CALL; }                           \
    case DT_FLOAT:  { typedef float          TT; CALL; }                           \
    case DT_DOUBLE: { typedef double         TT; CALL; }                           \
    default: return WRONG_PARAM;                                                   \
  }

static bool masksArgOk(int nMasks, int nBands, const void* pValidBytes)
{
  return (nMasks == 0 || nMasks == 1 || nMasks == nBands) && !(nMasks > 0 && !pValidBytes);
}

extern "C" {

lerc_status lerc_computeCompressedSize_4D(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
  int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes,
  const unsigned char* pUsesNoData, const double* noDataValues)
{
  if (!numBytes) return WRONG_PARAM;
  *numBytes = 0;
  if (!pData || dataType >= DT_UNDEF || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0) return WRONG_PARAM;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return WRONG_PARAM;
  unsigned written = 0;
  ORC_DISPATCH(dataType,
    if (!dimsOk(nDepth, nCols, nRows, sizeof(TT))) return DIMS_TOO_LARGE;
    return encodeBands((const TT*)pData, -1, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, *numBytes,
      nullptr, 0, written, pUsesNoData, noDataValues))
}

lerc_status lerc_encode_4D(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows, int nBands,
  int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer, unsigned int outBufferSize,
  unsigned int* nBytesWritten, const unsigned char* pUsesNoData, const double* noDataValues)
{
  if (!nBytesWritten) return WRONG_PARAM;
  *nBytesWritten = 0;
  if (!pData || dataType >= DT_UNDEF || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0
    || !pOutBuffer || !outBufferSize)
    return WRONG_PARAM;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return WRONG_PARAM;
  unsigned needed = 0;
  ORC_DISPATCH(dataType,
    if (!dimsOk(nDepth, nCols, nRows, sizeof(TT))) return DIMS_TOO_LARGE;
    memset(pOutBuffer, 0, outBufferSize);
    return encodeBands((const TT*)pData, -1, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, needed,
      pOutBuffer, outBufferSize, *nBytesWritten, pUsesNoData, noDataValues))
}

lerc_status lerc_computeCompressedSizeForVersion(const void* pData, int codecVersion, unsigned int dataType, int nDepth,
  int nCols, int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes)
{
  if (!numBytes) return WRONG_PARAM;
  *numBytes = 0;
  if (codecVersion >= 0 && codecVersion <= 5)    // Lerc.cpp:339-347
  {
    if (!pData || dataType >= DT_UNDEF || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0) return WRONG_PARAM;
    if (!masksArgOk(nMasks, nBands, pValidBytes)) return WRONG_PARAM;
    unsigned written = 0;
    ORC_DISPATCH(dataType,
      if (!dimsOk(nDepth, nCols, nRows, sizeof(TT))) return DIMS_TOO_LARGE;
      return encodeBandsOld((const TT*)pData, codecVersion, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, *numBytes,
        nullptr, 0, written))
  }
  if (codecVersion > kCurrentVersion) return WRONG_PARAM;
  return lerc_computeCompressedSize_4D(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr,
    numBytes, nullptr, nullptr);
}

lerc_status lerc_encodeForVersion(const void* pData, int codecVersion, unsigned int dataType, int nDepth, int nCols,
  int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer,
  unsigned int outBufferSize, unsigned int* nBytesWritten)
{
  if (!nBytesWritten) return WRONG_PARAM;
  *nBytesWritten = 0;
  if (codecVersion >= 0 && codecVersion <= 5)    // Lerc.cpp:378-386
  {
    if (!pData || dataType >= DT_UNDEF || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0
      || !pOutBuffer || !outBufferSize)
      return WRONG_PARAM;
    if (!masksArgOk(nMasks, nBands, pValidBytes)) return WRONG_PARAM;
    unsigned needed = 0;
    ORC_DISPATCH(dataType,
      if (!dimsOk(nDepth, nCols, nRows, sizeof(TT))) return DIMS_TOO_LARGE;
      memset(pOutBuffer, 0, outBufferSize);
      return encodeBandsOld((const TT*)pData, codecVersion, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, needed,
        pOutBuffer, outBufferSize, *nBytesWritten))
  }
  if (codecVersion > kCurrentVersion) return WRONG_PARAM;
  return lerc_encode_4D(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, pOutBuffer,
    outBufferSize, nBytesWritten, nullptr, nullptr);
}

lerc_status lerc_computeCompressedSize(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
  int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes)
{
  return lerc_computeCompressedSizeForVersion(pData, -1, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes,
    maxZErr, numBytes);
}

lerc_status lerc_encode(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows, int nBands,
  int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer, unsigned int outBufferSize,
  unsigned int* nBytesWritten)
{
  return lerc_encodeForVersion(pData, -1, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr,
    pOutBuffer, outBufferSize, nBytesWritten);
}

lerc_status lerc_getBlobInfo(const unsigned char* pLercBlob, unsigned int blobSize, unsigned int* infoArray,
  double* dataRangeArray, int infoArraySize, int dataRangeArraySize)
{
  if (!pLercBlob || !blobSize || (!infoArray && !dataRangeArray) || ((infoArraySize <= 0) && (dataRangeArraySize <= 0)))
    return WRONG_PARAM;
  Info li;
  Err e = getInfo(pLercBlob, blobSize, li);
  if (e != OK) return e;
  if (infoArray)
  {
    const unsigned v[11] = { (unsigned)li.version, (unsigned)li.dt, (unsigned)li.nDepth, (unsigned)li.nCols,
      (unsigned)li.nRows, (unsigned)li.nBands, (unsigned)li.numValid, li.blobSize, (unsigned)li.nMasks,
      (unsigned)li.nDepth, (unsigned)li.nUsesNoData };
    if (infoArraySize > 0) memset(infoArray, 0, infoArraySize * sizeof(unsigned));
    for (int i = 0; i < infoArraySize && i < 11; i++) infoArray[i] = v[i];
  }
  if (dataRangeArray)
  {
    if (dataRangeArraySize > 0) memset(dataRangeArray, 0, dataRangeArraySize * sizeof(double));
    const bool nd = (li.nDepth > 1) && (li.nUsesNoData > 0);
    const double v[3] = { !nd ? li.zMin : -1, !nd ? li.zMax : -1, li.maxZErr };
    for (int i = 0; i < dataRangeArraySize && i < 3; i++) dataRangeArray[i] = v[i];
  }
  return OK;
}

lerc_status lerc_getDataRanges(const unsigned char* pLercBlob, unsigned int blobSize, int nDepth, int nBands,
  double* pMins, double* pMaxs)
{
  if (!pLercBlob || !blobSize || !pMins || !pMaxs || nDepth <= 0 || nBands <= 0) return WRONG_PARAM;
  Info li;
  return getInfo(pLercBlob, blobSize, li, pMins, pMaxs, (size_t)nDepth * (size_t)nBands);
}

lerc_status lerc_decode_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks, unsigned char* pValidBytes,
  int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData, unsigned char* pUsesNoData,
  double* noDataValues)
{
  if (!pLercBlob || !blobSize || !pData || dataType >= DT_UNDEF || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0)
    return WRONG_PARAM;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return WRONG_PARAM;
  ORC_DISPATCH(dataType,
    return decodeBands((TT*)pData, pLercBlob, blobSize, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, pUsesNoData,
      noDataValues))
}

lerc_status lerc_decode(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks, unsigned char* pValidBytes,
  int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData)
{
  return lerc_decode_4D(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, dataType, pData, nullptr, nullptr);
}

lerc_status lerc_decodeToDouble_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
  unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData, unsigned char* pUsesNoData,
  double* noDataValues)
{
  if (!pLercBlob || !blobSize || !pData || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0) return WRONG_PARAM;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return WRONG_PARAM;
  Info li;
  Err e = getInfo(pLercBlob, blobSize, li);
  if (e != OK) return e;
  if (li.nDepth != nDepth || li.nCols != nCols || li.nRows != nRows || li.nBands != nBands) return FAILED;
  const int dt = li.dt;
  const size_t nVals = (size_t)nDepth * nCols * nRows * nBands;
  if (dt == DT_DOUBLE)
    return lerc_decode_4D(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, dt, pData, pUsesNoData, noDataValues);
  // decode into the tail of the caller's buffer, then widen in place front to back (Lerc_c_api_impl.cpp:288-300)
  void* tail = (u8*)pData + nVals * (sizeof(double) - dtSize(dt));
  lerc_status rc = lerc_decode_4D(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, dt, tail, pUsesNoData, noDataValues);
  if (rc != OK) return rc;
  ORC_DISPATCH(dt, { const TT* src = (const TT*)tail; for (size_t k = 0; k < nVals; k++) pData[k] = (double)src[k]; return OK; })
}

lerc_status lerc_decodeToDouble(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
  unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData)
{
  return lerc_decodeToDouble_4D(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, pData, nullptr, nullptr);
}

unsigned int orc_fletcher32(const unsigned char* bytes, int len) { return fletcher32(bytes, len); }

long long orc_blockTable(const unsigned char* blob, unsigned int blobSize, unsigned int* offsets, unsigned char* flags,
  long long capacity)
{
  Reader r{ blob, blobSize };
  Band b;
  if (!readHeader(r, b.hd)) return -1;
  if (!b.readMask(r)) return -2;
  const Header& h = b.hd;
  if (h.numValid == 0 || h.zMin == h.zMax) return -3;
  const size_t sz = dtSize(h.dt);
  if (h.version >= 4)
  {
    if (r.left < 2 * sz * h.nDepth) return -4;
    if (0 == memcmp(r.p, r.p + sz * h.nDepth, sz * h.nDepth)) return -3;
    r.skip(2 * sz * h.nDepth);
  }
  u8 sweep;
  if (!r.get(&sweep, 1) || sweep) return -5;
  if (h.tryHuffmanInt() || h.tryHuffmanFlt()) { u8 f; if (!r.get(&f, 1) || f != IEM_TILING) return -6; }
  const int mb = h.mbSize, nTV = (h.nRows + mb - 1) / mb, nTH = (h.nCols + mb - 1) / mb;
  long long nBlocks = 0;
  for (int it = 0; it < nTV; it++)
    for (int jt = 0; jt < nTH; jt++)
      for (int iD = 0; iD < h.nDepth; iD++)
      {
        int i0 = it * mb, i1 = std::min(h.nRows, i0 + mb), j0 = jt * mb, j1 = std::min(h.nCols, j0 + mb);
        int nValid = 0;
        for (int i = i0; i < i1; i++) for (int j = j0; j < j1; j++) nValid += b.mask.valid((int64_t)i * h.nCols + j);
        if (r.left < 1) return -7;
        const u8 flag = *r.p;
        if (nBlocks < capacity) { if (offsets) offsets[nBlocks] = (unsigned)(r.p - blob); if (flags) flags[nBlocks] = flag; }
        nBlocks++;
        r.skip(1);
        const int mode = flag & 3, tc = flag >> 6;
        const bool diff = h.version >= 5 && (flag & 4);
        if (mode == 2) continue;
        if (mode == 0) { if (!r.skip((size_t)nValid * sz)) return -8; continue; }
        int dtU = typeUsed((diff && h.dt < DT_FLOAT) ? DT_INT : h.dt, tc);
        if (!r.skip(dtSize(dtU))) return -9;
        if (mode == 3) continue;
        std::vector<unsigned> tmp;
        if (!decodeBitStuffer(r, tmp, (size_t)(i1 - i0) * (j1 - j0), h.version)) return -10;
      }
  return nBlocks;
}

}    // extern "C"
