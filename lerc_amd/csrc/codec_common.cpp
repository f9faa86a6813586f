// codec_common.cpp -- wire-format pieces that stay on the host (a few hundred bytes per band):
// header (Lerc2.cpp:724-917), mask RLE (RLE.cpp:32-331), blob info walk (Lerc.cpp:92-182), and the
// per-thread device context.
#include "codec.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <future>
#include <vector>

namespace lerc {

// ------------------------------------------------------------------------------------------------
// header
// ------------------------------------------------------------------------------------------------
u32 headerBytes(int v)
{
  return 6 + 4 + (v >= 3 ? 4 : 0) + 4 * (v >= 4 ? 7 : 6) + (v >= 6 ? 8 : 0) + 8 * (v >= 6 ? 5 : 3);
}

namespace {
struct Out
{
  u8* p;
  template<class T> void put(const T& v) { memcpy(p, &v, sizeof(T)); p += sizeof(T); }
};
struct In
{
  const u8* p;
  size_t left;
  template<class T> bool get(T& v) { if (left < sizeof(T)) return false; memcpy(&v, p, sizeof(T)); p += sizeof(T); left -= sizeof(T); return true; }
};
}

void writeHeader(u8* dst, const Header& h)
{
  Out o{ dst };
  memcpy(o.p, "Lerc2 ", 6); o.p += 6;
  o.put(h.version);
  if (h.version >= 3) o.put((u32)0);    // checksum slot, patched after the payload exists
  o.put(h.nRows); o.put(h.nCols);
  if (h.version >= 4) o.put(h.nDepth);
  o.put(h.numValid); o.put(h.mbSize); o.put(h.blobSize); o.put(h.dt);
  if (h.version >= 6) { o.put(h.nBlobsMore); o.put(h.passNoData); o.put(h.isInt); o.put(h.rsv3); o.put(h.rsv4); }
  o.put(h.maxZErr); o.put(h.zMin); o.put(h.zMax);
  if (h.version >= 6) { o.put(h.noDataVal); o.put(h.noDataValOrig); }
}

bool readHeader(const u8* src, size_t n, Header& h, size_t& used)
{
  h = Header();
  In in{ src, n };
  if (n < 6 || memcmp(src, "Lerc2 ", 6)) return false;
  in.p += 6; in.left -= 6;
  if (!in.get(h.version) || h.version < 0 || h.version > kCodecVersion) return false;
  if (h.version >= 3 && !in.get(h.checksum)) return false;
  h.nDepth = 1;
  bool ok = in.get(h.nRows) && in.get(h.nCols);
  if (ok && h.version >= 4) ok = in.get(h.nDepth);
  ok = ok && in.get(h.numValid) && in.get(h.mbSize) && in.get(h.blobSize) && in.get(h.dt);
  if (ok && h.version >= 6) ok = in.get(h.nBlobsMore) && in.get(h.passNoData) && in.get(h.isInt) && in.get(h.rsv3) && in.get(h.rsv4);
  ok = ok && in.get(h.maxZErr) && in.get(h.zMin) && in.get(h.zMax);
  if (ok && h.version >= 6) ok = in.get(h.noDataVal) && in.get(h.noDataValOrig);
  if (!ok) return false;
  if (h.nRows <= 0 || h.nCols <= 0 || h.nDepth <= 0 || h.numValid < 0 || h.mbSize <= 0 || h.blobSize <= 0
    || h.dt < DT_Char || h.dt > DT_Double)
    return false;
  const u64 nPix = (u64)h.nRows * h.nCols, lim = (u64)INT_MAX, bpp = (u64)dtSize(h.dt);
  if (nPix > lim || (u64)h.numValid > nPix) return false;
  if (h.mbSize > 32 || bpp * h.nDepth > lim || bpp * h.nDepth * nPix > lim) return false;
  used = (size_t)(in.p - src);
  return true;
}

// ------------------------------------------------------------------------------------------------
// RLE of the mask bytes.  Stream: [int16 n][payload] ...; n > 0: n literal bytes; n < 0: one byte
// repeated -n times; -32768 ends the stream.  A run is opened only where at least 5 equal bytes
// start and one more byte follows (RLE.cpp:166-172, RLE.h:45); segments are cut at 32767.
// ------------------------------------------------------------------------------------------------
// bytes [begin, end) of b[0 .. n), appended to out without the end marker.  `end` is n, or the start of a run that follows a
// run of five or more equal bytes -- so is `begin` (or 0): at such a place RLE.cpp's encoder has just closed a run and starts a
// literal stretch with an empty count, whatever came before; what it writes from there on hangs on nothing in front of it.
static void rleEncodeRange(const u8* b, size_t n, size_t begin, size_t end, std::vector<u8>& out)
{
  auto count = [&](int v) { short s = (short)v; out.push_back((u8)(s & 0xff)); out.push_back((u8)((s >> 8) & 0xff)); };
  // eight bytes at a time (a validity mask of a large raster is megabytes): five equal bytes from i on <=> the low
  // 32 bits of x ^ (x >> 8) are zero, x = the 8 bytes at i
  auto load8 = [&](size_t at) { u64 x; memcpy(&x, b + at, 8); return x; };
  size_t i = begin;
  while (i < end)
  {
    const size_t litBeg = i;
    while (i < end)
    {
      bool runStarts;
      if (i + 8 <= n) { const u64 x = load8(i); runStarts = (i + 5 < n) && (u32)(x ^ (x >> 8)) == 0u; }
      else runStarts = (i + 5 < n) && b[i] == b[i + 1] && b[i] == b[i + 2] && b[i] == b[i + 3] && b[i] == b[i + 4];
      if (runStarts) break;
      i++;
    }
    for (size_t at = litBeg; at < i;)
    {
      const size_t len = std::min<size_t>(32767, i - at);
      count((int)len);
      out.insert(out.end(), b + at, b + at + len);
      at += len;
    }
    if (i >= end) break;
    size_t t = i;
    const u64 same = 0x0101010101010101ull * b[i];
    while (t + 9 <= n && load8(t + 1) == same) t += 8;    // b[t + 1 .. t + 8] all equal b[i]
    while (t + 1 < n && b[t + 1] == b[i]) t++;
    for (size_t left = t - i + 1; left > 0;)
    {
      const size_t len = std::min<size_t>(32767, left);
      count(-(int)len);
      out.push_back(b[i]);
      left -= len;
    }
    i = t + 1;
  }
}

// A large mask (megabytes of bits) is cut where a long run ends -- see rleEncodeRange -- and the pieces are coded by a few
// threads: the coding of 8 MB takes one core a millisecond (most of it reading what the copy engine has just written), and with
// the block stream of a masked band down to a fifth of that (the one-launch encoder) it was what an encode call waited for.
// Piece p of K begins at the first place in its window [p n / K, (p + 1) n / K) where a run begins right behind five equal
// bytes -- a piece whose window holds no such place does not exist, the piece in front of it runs on -- so every worker finds
// its own range without asking anybody.
static size_t rleCutIn(const u8* b, size_t n, size_t K, size_t p)    // 0: none (p >= 1)
{
  const size_t from = std::max<size_t>(p * (n / K), 6), to = std::min(n, (p + 1) * (n / K));
  for (size_t i = from; i < to; i++)
    if (b[i] != b[i - 1] && b[i - 1] == b[i - 2] && b[i - 1] == b[i - 3] && b[i - 1] == b[i - 4] && b[i - 1] == b[i - 5]) return i;
  return 0;
}
static void rlePiece(const u8* b, size_t n, size_t K, size_t p, std::vector<u8>& out)
{
  const size_t begin = p ? rleCutIn(b, n, K, p) : 0;
  if (p && !begin) return;
  // (where this piece ends: the first cut of a window behind it.  A mask without runs has no cut anywhere: each window is then
  // looked through once here and once by its own worker, which gives up -- bytewise, but without coding anything)
  size_t end = n;
  for (size_t q = p + 1; q < K; q++) { const size_t c = rleCutIn(b, n, K, q); if (c) { end = c; break; } }
  // (into a vector of this thread's own and handed over at the end: the pieces' vectors lie side by side, and every push_back
  // writes its vector's header -- eight threads on one cache line ran four times slower than one)
  std::vector<u8> mine;
  mine.reserve((end - begin) / 8 + 64);
  rleEncodeRange(b, n, begin, end, mine);
  out.swap(mine);
}
static size_t rlePieces(size_t n)
{
  // (LERC_AMD_RLE_PIECE=<bytes>, LERC_AMD_RLE_THREADS=<n>: test knobs, so that small masks are cut too)
  static const size_t kPiece = []() -> size_t { const char* e = getenv("LERC_AMD_RLE_PIECE"); const long v = e ? atol(e) : 0; return v >= 16 ? (size_t)v : (size_t)(512 << 10); }();
  static const size_t kMaxPieces = []() -> size_t { const char* e = getenv("LERC_AMD_RLE_THREADS"); const long v = e ? atol(e) : 0; return v >= 1 ? (size_t)v : (size_t)8; }();
  return std::max<size_t>(1, std::min<size_t>(kMaxPieces, n / kPiece));
}

// ready(): called by every worker before it looks at a byte (the bits may still be on their way from the device: the workers
// are started while they travel, so that starting them costs the caller nothing); false: give up, out stays empty
bool rleEncodeWhenReady(const u8* b, size_t n, const std::function<bool()>& ready, std::vector<u8>& out)
{
  out.clear();
  const size_t K = rlePieces(n);
  std::vector<std::vector<u8> > part(K);
  // ready() is the caller's alone (a HIP event wait: eight threads waiting for one event took turns, a tenth of a millisecond
  // each); the workers watch a flag
  std::atomic<int> state(0);    // 1: the bytes are there, -1: they will not come
  std::vector<std::future<void> > job;
  std::vector<size_t> inLine;
  job.reserve(K); inLine.reserve(K);
  // (whatever throws before the flag is set -- ready() itself, say --: the workers must not be left spinning on it, or the
  // futures' destructors would wait for them for ever; declared behind the futures, so it goes first)
  struct Release { std::atomic<int>& s; ~Release() { int zero = 0; s.compare_exchange_strong(zero, -1, std::memory_order_acq_rel); } } release{ state };
  for (size_t p = 1; p < K; p++)
  {
    try
    {
      job.push_back(std::async(std::launch::async, [&, p]()
      {
        int st;
        for (unsigned spins = 0; (st = state.load(std::memory_order_acquire)) == 0; spins++)    // (the bits travel for a third of a millisecond: yield first, then sleep in short steps)
          if (spins < 64) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (st > 0) rlePiece(b, n, K, p, part[p]);
      }));
    }
    catch (...) { inLine.push_back(p); }    // (no thread to be had)
  }
  static const bool kTL = getenv("LERC_AMD_HOST_TIMES") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
  const bool good = ready();
  const double tReady = us();
  state.store(good ? 1 : -1, std::memory_order_release);
  if (good)
  {
    rlePiece(b, n, K, 0, part[0]);
    for (size_t p : inLine) rlePiece(b, n, K, p, part[p]);
  }
  const double tMine = us();
  for (auto& j : job) j.get();
  if (kTL) fprintf(stderr, "  [tl] rle: %zu pieces, bytes there after %.1f us, own piece +%.1f, all pieces +%.1f\n", K, tReady, tMine - tReady, us() - tReady);
  if (!good) return false;
  size_t total = 2;
  for (auto& v : part) total += v.size();
  out.reserve(total);
  for (auto& v : part) out.insert(out.end(), v.begin(), v.end());
  out.push_back(0x00); out.push_back(0x80);    // -32768
  return true;
}

void rleEncode(const u8* b, size_t n, std::vector<u8>& out)
{
  rleEncodeWhenReady(b, n, []() { return true; }, out);
}

bool rleDecode(const u8* src, size_t left, u8* dst, size_t dstSize, size_t* written)
{
  if (written) *written = 0;
  if (!src || !dst || left < 2) return false;
  size_t at = 0;
  for (;;)
  {
    if (left < 2) return false;
    const short cnt = (short)(src[0] | (src[1] << 8));
    src += 2; left -= 2;
    if (cnt == -32768) { if (written) *written = at; return true; }
    const size_t n = (size_t)(cnt < 0 ? -cnt : cnt), payload = cnt > 0 ? n : 1;
    if (left < payload + 2 || at + n > dstSize) return false;    // + 2: a count always follows (RLE.cpp:310)
    if (cnt > 0) memcpy(dst + at, src, n); else memset(dst + at, src[0], n);
    at += n; src += payload; left -= payload;
  }
}

// ------------------------------------------------------------------------------------------------
// blob info: hop over concatenated band blobs (Lerc.cpp:92-182, :1012-1042)
// ------------------------------------------------------------------------------------------------
static bool peekBand(const u8* p, size_t n, Header& h, bool& hasMask, size_t& hdrLen)
{
  if (!p || !readHeader(p, n, h, hdrLen)) return false;
  int nm = 0;
  if (n - hdrLen < 4) return false;
  memcpy(&nm, p + hdrLen, 4);
  if (nm < 0) return false;
  hasMask = nm > 0;
  return true;
}

// per-depth ranges of one band (Lerc2::GetRanges, Lerc2.cpp:516-573); host-side header parsing only
static u32 bandRanges(const u8* p, size_t n, int iBand, const Header& h, size_t hdrLen, double* mins, double* maxs, size_t nElem)
{
  const int nD = h.nDepth;
  if (nD <= 0 || iBand < 0 || !mins || !maxs) return kWrongParam;
  if (nElem < ((size_t)iBand + 1) * (size_t)nD) return kBufferTooSmall;
  if (nD == 1) { mins[iBand] = h.zMin; maxs[iBand] = h.zMax; return kOk; }
  if (h.passNoData) return kHasNoData;
  if (h.version < 4) return kFailed;
  double* lo = mins + (size_t)iBand * nD;
  double* hi = maxs + (size_t)iBand * nD;
  if (h.numValid == 0) { for (int m = 0; m < nD; m++) lo[m] = hi[m] = 0; return kOk; }
  if (h.zMin == h.zMax) { for (int m = 0; m < nD; m++) lo[m] = hi[m] = h.zMin; return kOk; }
  int nm = 0;
  memcpy(&nm, p + hdrLen, 4);
  const size_t at = hdrLen + 4 + (size_t)nm, sz = (size_t)dtSize(h.dt);
  if (nm < 0 || n < at + 2 * sz * nD) return kFailed;
  for (int m = 0; m < nD; m++)
  {
    lo[m] = typedFromBits(getBytes(p + at + m * sz, (int)sz), h.dt);
    hi[m] = typedFromBits(getBytes(p + at + (nD + m) * sz, (int)sz), h.dt);
  }
  return kOk;
}

u32 getBlobInfo(const u8* blob, u32 n, BlobInfo& info, double* mins, double* maxs, size_t nElem)
{
  info = BlobInfo();
  Header h;
  bool hasMask = false;
  size_t hdrLen = 0;
  int nMasks = 0;
  if (!peekBand(blob, n, h, hasMask, hdrLen)) return kFailed;    // Lerc1 legacy blobs are not handled by this library
  info.version = h.version; info.nDepth = h.nDepth; info.nCols = h.nCols; info.nRows = h.nRows;
  info.numValid = h.numValid; info.blobSize = (u32)h.blobSize; info.dt = h.dt;
  info.zMin = h.zMin; info.zMax = h.zMax; info.maxZErr = h.maxZErr; info.nUsesNoData = h.passNoData ? 1 : 0;
  bool more = (h.version <= 5) || (h.nBlobsMore > 0);
  if (hasMask || info.numValid == 0) nMasks = 1;
  if (mins && maxs) { const u32 e = bandRanges(blob, n, 0, h, hdrLen, mins, maxs, nElem); if (e != kOk) return e; }
  info.nBands = 1;
  if (info.blobSize > n) return kFailed;
  Header hn;
  while (more && peekBand(blob + info.blobSize, n - info.blobSize, hn, hasMask, hdrLen))
  {
    if (hn.nDepth != info.nDepth || hn.nCols != info.nCols || hn.nRows != info.nRows || hn.dt != info.dt) return kFailed;
    more = (hn.version <= 5) || (hn.nBlobsMore > 0);
    if (hn.passNoData) info.nUsesNoData++;
    if (hasMask || hn.numValid != info.numValid) nMasks = 2;
    if ((size_t)info.blobSize > (size_t)UINT_MAX - (size_t)hn.blobSize) return kFailed;
    if ((size_t)info.blobSize + (size_t)hn.blobSize > (size_t)n) return kFailed;
    info.zMin = std::min(info.zMin, hn.zMin);
    info.zMax = std::max(info.zMax, hn.zMax);
    info.maxZErr = std::max(info.maxZErr, hn.maxZErr);
    if (mins && maxs)
    {
      const u32 e = bandRanges(blob + info.blobSize, n - info.blobSize, info.nBands, hn, hdrLen, mins, maxs, nElem);
      if (e != kOk) return e;
    }
    info.blobSize += (u32)hn.blobSize;
    info.nBands++;
  }
  info.nMasks = nMasks > 1 ? info.nBands : nMasks;
  if (info.nUsesNoData > 0) info.nUsesNoData = info.nBands;
  return kOk;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
Context::Context()
{
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
  {
    lastError = "lerc_amd: no HIP device available -- this library has no CPU path";
    fprintf(stderr, "%s\n", lastError.c_str());
    return;
  }
  if (hipStreamCreateWithFlags(&m_stream, hipStreamNonBlocking) != hipSuccess)
  {
    lastError = "lerc_amd: hipStreamCreate failed";
    fprintf(stderr, "%s\n", lastError.c_str());
    return;
  }
  m_ok = true;
}

Context::~Context()
{
  if (m_slab) hipFree(m_slab);
  if (m_pinned) hipHostFree(m_pinned);
  if (m_pinnedAux) hipHostFree(m_pinnedAux);
  if (m_asyncPinned) hipHostFree(m_asyncPinned);
  for (u8* st : m_state) if (st) hipFree(st);
  if (m_auxEvent) hipEventDestroy(m_auxEvent);
  if (m_forkEvent) hipEventDestroy(m_forkEvent);
  if (m_sideStream) { hipStreamSynchronize(m_sideStream); hipStreamDestroy(m_sideStream); }
  for (hipEvent_t e : m_eventPool) hipEventDestroy(e);
  if (m_stream) hipStreamDestroy(m_stream);
}

bool Context::reserve(size_t bytes)
{
  if (m_sideUsed) { m_sideUsed = false; hipStreamSynchronize(m_sideStream); }    // (a call that ended early: its side copy reads the scratch)
  m_used = 0;
  if (bytes <= m_cap) { poisonScratch(bytes); return true; }
  if (m_slab) { hipStreamSynchronize(activeStream()); hipFree(m_slab); m_slab = nullptr; m_cap = 0; }
  const size_t want = bytes + bytes / 8 + (1u << 20);
  if (hipMalloc((void**)&m_slab, want) != hipSuccess) { lastError = "lerc_amd: hipMalloc failed"; return false; }
  m_cap = want;
  poisonScratch(bytes);
  return true;
}

// Test aid: LERC_AMD_POISON=<byte> fills the scratch a call is about to use with that byte, so that a kernel reading what
// nobody wrote (block offsets of a damaged blob, say) meets 0xA5A5A5A5 or 0xFFFFFFFF instead of the zeros of a fresh
// allocation or the plausible leftovers of the call before (tests/test_sim_kernels.py runs the damaged-blob cases that way).
void Context::poisonScratch(size_t bytes)
{
  static const char* env = getenv("LERC_AMD_POISON");
  if (!env || !m_slab) return;
  hipMemsetAsync(m_slab, (int)strtol(env, nullptr, 0) & 255, bytes < m_cap ? bytes : m_cap, activeStream());
}

void* Context::alloc(size_t bytes, size_t align)
{
  size_t at = (m_used + align - 1) / align * align;
  if (at + bytes > m_cap) return nullptr;
  m_used = at + bytes;
  return m_slab + at;
}

void* Context::pinned(size_t bytes)
{
  if (bytes <= m_pinnedCap) return m_pinned;
  if (m_pinned) hipHostFree(m_pinned);
  m_pinned = nullptr; m_pinnedCap = 0;
  if (hipHostMalloc(&m_pinned, bytes + 4096, hipHostMallocDefault) != hipSuccess) return nullptr;
  m_pinnedCap = bytes + 4096;
  return m_pinned;
}

u8* Context::persistentState(int area, size_t bytes)
{
  // (the cells of area 1 carry the call's 32-bit epoch: long before a tag can come round again, wipe them)
  const bool wipe = area == 1 && (++m_stateCalls & ((1ull << 30) - 1)) == 0;
  if (bytes <= m_stateCap[area] && !wipe) return m_state[area];
  hipStreamSynchronize(activeStream());
  if (bytes > m_stateCap[area])
  {
    if (m_state[area]) hipFree(m_state[area]);
    m_state[area] = nullptr; m_stateCap[area] = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc((void**)&m_state[area], want) != hipSuccess) { lastError = "lerc_amd: hipMalloc failed"; return nullptr; }
    m_stateCap[area] = want;
  }
  // (ON the context's stream: it is a non-blocking one, which a hipMemset on the default stream is not ordered with)
  if (hipMemsetAsync(m_state[area], 0, m_stateCap[area], activeStream()) != hipSuccess || hipStreamSynchronize(activeStream()) != hipSuccess) return nullptr;
  return m_state[area];
}

void Context::wipePersistentState()
{
  hipStreamSynchronize(activeStream());
  for (int a = 0; a < 2; a++) if (m_state[a]) hipMemsetAsync(m_state[a], 0, m_stateCap[a], activeStream());    // (see persistentState)
  hipStreamSynchronize(activeStream());
}

u8* Context::asyncSlot(unsigned ticket)
{
  if (!m_asyncPinned && hipHostMalloc((void**)&m_asyncPinned, (size_t)kAsyncSlots * kAsyncSlotBytes, hipHostMallocDefault) != hipSuccess)
    m_asyncPinned = nullptr;
  return m_asyncPinned ? m_asyncPinned + (size_t)(ticket % kAsyncSlots) * kAsyncSlotBytes : nullptr;
}

void* Context::pinnedAux(size_t bytes)
{
  if (bytes <= m_pinnedAuxCap) return m_pinnedAux;
  if (m_pinnedAux) hipHostFree(m_pinnedAux);
  m_pinnedAux = nullptr; m_pinnedAuxCap = 0;
  if (hipHostMalloc(&m_pinnedAux, bytes + 4096, hipHostMallocDefault) != hipSuccess) return nullptr;
  m_pinnedAuxCap = bytes + 4096;
  return m_pinnedAux;
}

hipStream_t Context::forkSide()
{
  if (!m_sideStream && hipStreamCreateWithFlags(&m_sideStream, hipStreamNonBlocking) != hipSuccess) { m_sideStream = nullptr; return nullptr; }
  if (!m_forkEvent && hipEventCreateWithFlags(&m_forkEvent, hipEventDisableTiming) != hipSuccess) { m_forkEvent = nullptr; return nullptr; }
  if (hipEventRecord(m_forkEvent, activeStream()) != hipSuccess || hipStreamWaitEvent(m_sideStream, m_forkEvent, 0) != hipSuccess) return nullptr;
  m_sideUsed = true;    // (sync() and reset() wait for it too from now on)
  return m_sideStream;
}

hipEvent_t Context::auxEvent()
{
  if (!m_auxEvent && hipEventCreateWithFlags(&m_auxEvent, hipEventDisableTiming) != hipSuccess) m_auxEvent = nullptr;
  return m_auxEvent;
}

// Waits for the stream.  A blocking hipStreamSynchronize wakes the thread up tens of microseconds late, which is
// a tenth of a whole 8192 x 8192 call; calls this short are worth polling for.
bool Context::sync()
{
  hipStream_t st = activeStream();
  if (m_sideUsed)    // (the copy beside the stream, forkSide(): nothing of the call may outlive the call)
  {
    m_sideUsed = false;
    if (hipStreamSynchronize(m_sideStream) != hipSuccess) return false;
  }
  for (int i = 0; i < 20000; i++)
  {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) return false;
  }
  return hipStreamSynchronize(st) == hipSuccess;
}

hipEvent_t Context::profEvent()
{
  if (!m_eventPool.empty()) { hipEvent_t e = m_eventPool.back(); m_eventPool.pop_back(); return e; }
  hipEvent_t e = nullptr;
#ifdef hipEventDisableSystemFence
  // a plain event record carries a system-scope release (cache write-back) and costs ~10 us between two kernels
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) == hipSuccess) return e;
#endif
  hipEventCreate(&e);
  return e;
}

// Consecutive groups share their boundary event: when a group begins right after the previous one ended (nothing
// enqueued in between by this library), the previous end event is the new begin event.
void Context::profBegin(const char* name)
{
  ProfEntry pe{ name, nullptr, nullptr, true, true };
  if (m_lastEndFresh && !m_pending.empty() && m_pending.back().b) { pe.a = m_pending.back().b; pe.ownsA = false; }
  else { pe.a = profEvent(); hipEventRecord(pe.a, activeStream()); }
  m_lastEndFresh = false;
  m_pending.push_back(pe);
}

void Context::profEnd()
{
  if (m_pending.empty()) return;
  ProfEntry& pe = m_pending.back();
  pe.b = profEvent();
  hipEventRecord(pe.b, activeStream());
  m_lastEndFresh = true;
}

void Context::profCollect()
{
  for (ProfEntry& pe : m_pending)
  {
    float ms = 0;
    if (pe.a && pe.b && hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess)
    {
      bool found = false;
      for (ProfAcc& a : m_acc) if (a.name == pe.name) { a.ms += ms; a.n++; found = true; break; }
      if (!found) m_acc.push_back(ProfAcc{ pe.name, ms, 1 });
    }
  }
  for (ProfEntry& pe : m_pending)
  {
    if (pe.a && pe.ownsA) m_eventPool.push_back(pe.a);
    if (pe.b) m_eventPool.push_back(pe.b);
  }
  m_pending.clear();
  m_lastEndFresh = false;
}

std::string Context::profReport(bool reset)
{
  profCollect();
  std::string out;
  char line[256];
  for (const ProfAcc& a : m_acc) { snprintf(line, sizeof(line), "%s %.6f %d\n", a.name.c_str(), a.ms, a.n); out += line; }
  if (reset) m_acc.clear();
  return out;
}

}    // namespace lerc
