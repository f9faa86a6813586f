// rle_kernels.hip -- the run-length coding of a band's validity bits (RLE::compress, RLE.cpp:123-254) on the device.
//
// Stream: [int16 n][payload] ...; n > 0: n literal bytes follow; n < 0: one byte follows, repeated -n times; -32768 ends the
// stream.  The reference scans front to back: in a literal stretch, at position i, a run is opened where at least 5 equal
// bytes start and one more byte follows (i + 5 < nBytes, RLE.cpp:166-172); the run then takes every byte equal to its first;
// literal stretches and runs are cut at 32767.  Said without the scan: the bytes fall into maximal sequences of equal bytes; a
// sequence of 5 or more is a run -- unless it is exactly the last 5 bytes of the array --, everything else is literal, and
// neighbouring literal sequences form one stretch.  Whether a byte is in a run is therefore a LOCAL question (5 bytes to
// either side answer it), and where its segment -- run or stretch -- begins and ends are two scans ("last segment head at or
// before", "first segment tail at or behind").  With both known, every byte knows what it writes: a literal byte itself,
// plus the two count bytes of its chunk if it is the chunk's first; a run byte the three bytes of a token if it is the first
// of 32767.  A prefix sum over those says where.
//
// A thread takes 16 consecutive bytes, so the scans run over nBytes / 16 values: five small launches around three passes
// over the bits -- 8 MB of them (an 8192 x 8192 mask) in a few tens of microseconds, instead of 0.34 ms on PCIe to the host
// and 0.29 ms in eight host threads.  The bytes are the host coder's (codec_common.cpp: rleEncode), which is pinned on the
// reference's; the tests compare the two on every kind of mask.
#include "kernels.h"
#include "wave_utils.h"

namespace lerc {

namespace {

constexpr u32 kSeg = 32767u;           // longest run / literal chunk a count can hold
constexpr u32 kNone = 0xFFFFFFFFu;

// The 48 bytes around a thread's 16: [i0 - 16, i0 + 32); bytes outside the array read as "different from everything" (the
// flags below never compare two of them with each other as equal, because indices outside the array are tested first).
struct Around
{
  u32 w[12];
  __device__ __forceinline__ u32 at(int k) const { return (w[(k + 16) >> 2] >> (8 * ((k + 16) & 3))) & 0xFFu; }    // k: relative to i0, -16 .. 31
};

__device__ __forceinline__ Around loadAround(const u8* __restrict__ b, u32 n, u32 i0)
{
  Around a;
#pragma unroll
  for (int q = 0; q < 3; q++)
  {
    const i64 at = (i64)i0 - 16 + 16 * q;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at >= 0 && (u64)at + 16 <= n) v = *reinterpret_cast<const uint4*>(b + at);
    else if (at < (i64)n && at + 16 > 0)
    {
      u32 t4[4] = { 0, 0, 0, 0 };
#pragma unroll
      for (int k = 0; k < 16; k++) if (at + k >= 0 && at + k < (i64)n) t4[k >> 2] |= (u32)b[at + k] << (8 * (k & 3));
      v = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    }
    a.w[4 * q] = v.x; a.w[4 * q + 1] = v.y; a.w[4 * q + 2] = v.z; a.w[4 * q + 3] = v.w;
  }
  return a;
}

// what the 18 positions i0 - 1 .. i0 + 16 are: bit k + 1 of `run` = position i0 + k lies in a run
struct Flags
{
  u32 run;      // bits 0 .. 17: positions i0 - 1 .. i0 + 16 (bits of positions outside the array: 0)
  u32 head;     // bits 0 .. 15: position i0 + k begins a segment (a run, or a stretch of literal bytes)
  u32 tail;     // ... ends one
};

__device__ __forceinline__ Flags classify(const Around& a, u32 n, u32 i0)
{
  Flags f;
  f.run = 0; f.head = 0; f.tail = 0;
  // eq bit k + 7: byte at i0 + k equals the byte before it (both inside the array); k = -6 .. 22
  u32 eq = 0;
#pragma unroll
  for (int k = -6; k <= 22; k++)
  {
    const i64 p = (i64)i0 + k;
    if (p >= 1 && p < (i64)n && a.at(k) == a.at(k - 1)) eq |= 1u << (k + 7);
  }
#pragma unroll
  for (int k = -1; k <= 16; k++)
  {
    const i64 p = (i64)i0 + k;
    if (p < 0 || p >= (i64)n) continue;
    // equal bytes to the left (at most 5 are looked at) and to the right
    u32 left = 0, right = 0;
#pragma unroll
    for (int d = 0; d < 5; d++) { if (left == (u32)d && ((eq >> (k - d + 7)) & 1u)) left++; }
#pragma unroll
    for (int d = 1; d <= 5; d++) { if (right == (u32)(d - 1) && ((eq >> (k + d + 7)) & 1u)) right++; }
    bool run = left + 1u + right >= 5u;
    // ... but a sequence that is exactly the array's last five bytes opens no run: no byte follows it (RLE.cpp:166-172)
    if (left < 5u && (u64)p - left + 5u >= (u64)n) run = false;
    if (run) f.run |= 1u << (k + 1);
  }
#pragma unroll
  for (int k = 0; k < 16; k++)
  {
    const i64 p = (i64)i0 + k;
    if (p >= (i64)n) continue;
    const bool run = (f.run >> (k + 1)) & 1u, runBefore = (f.run >> k) & 1u, runBehind = (f.run >> (k + 2)) & 1u;
    const bool sameBefore = (eq >> (k + 7)) & 1u, sameBehind = (eq >> (k + 8)) & 1u;
    if (p == 0 || run != runBefore || (run && !sameBefore)) f.head |= 1u << k;
    if (p == (i64)n - 1 || run != runBehind || (run && !sameBehind)) f.tail |= 1u << k;
  }
  return f;
}

// pass 1: per thread, the last segment head and the first segment tail among its 16 positions
__global__ void __launch_bounds__(256) k_rle_flags(const u8* __restrict__ b, u32 n, u32 nThreads, u32* __restrict__ lastHead, u32* __restrict__ firstTail)
{
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nThreads) return;
  const u32 i0 = 16u * t;
  const Flags f = classify(loadAround(b, n, i0), n, i0);
  lastHead[t] = f.head ? i0 + (31u - (u32)__clz((int)f.head)) + 1u : 0u;    // position + 1; 0: none (the scan takes the maximum)
  firstTail[t] = f.tail ? i0 + (u32)(__ffs((int)f.tail) - 1) : kNone;      // (the scan takes the minimum)
}

// inclusive scans over the threads' values: forward maximum / backward minimum, 1024 values per workgroup
template<bool MAXFWD>
__device__ __forceinline__ u32 op(u32 a, u32 c) { return MAXFWD ? (a > c ? a : c) : (a < c ? a : c); }

template<bool MAXFWD>
__global__ void __launch_bounds__(256) k_rle_scan_local(u32* __restrict__ v, u32 n, u32* __restrict__ partial)
{
  __shared__ u32 s_w[4];
  const u32 ident = MAXFWD ? 0u : kNone;
  const u32 base = blockIdx.x * 1024u + threadIdx.x * 4u;    // in scan order
  u32 a[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
  {
    const u32 e = base + i;
    a[i] = e < n ? v[MAXFWD ? e : n - 1u - e] : ident;
    if (i) a[i] = op<MAXFWD>(a[i - 1], a[i]);
  }
  u32 inc = a[3];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (laneId() >= d) inc = op<MAXFWD>(inc, o); }
  if (laneId() == 63) s_w[waveId()] = inc;
  __syncthreads();
  u32 before = __shfl_up(inc, 1u);
  if (laneId() == 0) before = ident;
  for (int i = 0; i < waveId(); i++) before = op<MAXFWD>(before, s_w[i]);
#pragma unroll
  for (int i = 0; i < 4; i++)
  {
    const u32 e = base + i;
    if (e < n) v[MAXFWD ? e : n - 1u - e] = op<MAXFWD>(before, a[i]);
  }
  if (threadIdx.x == 255) partial[blockIdx.x] = op<MAXFWD>(before, a[3]);
}

template<bool MAXFWD>
__global__ void __launch_bounds__(256) k_rle_scan_partials(u32* __restrict__ partial, u32 nPartials)
{
  __shared__ u32 s_w[4];
  const u32 ident = MAXFWD ? 0u : kNone;
  u32 carry = ident;
  for (u32 b0 = 0; b0 < nPartials; b0 += 256)
  {
    const u32 i = b0 + threadIdx.x;
    const u32 mine = i < nPartials ? partial[i] : ident;
    u32 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (laneId() >= d) inc = op<MAXFWD>(inc, o); }
    __syncthreads();
    if (laneId() == 63) s_w[waveId()] = inc;
    __syncthreads();
    u32 before = __shfl_up(inc, 1u);
    if (laneId() == 0) before = ident;
    for (int k = 0; k < waveId(); k++) before = op<MAXFWD>(before, s_w[k]);
    if (i < nPartials) partial[i] = op<MAXFWD>(carry, before);    // what lies in front of workgroup i (exclusive)
    carry = op<MAXFWD>(carry, op<MAXFWD>(op<MAXFWD>(s_w[0], s_w[1]), op<MAXFWD>(s_w[2], s_w[3])));
  }
}

template<bool MAXFWD>
__global__ void __launch_bounds__(256) k_rle_scan_add(u32* __restrict__ v, u32 n, const u32* __restrict__ partial)
{
  const u32 add = partial[blockIdx.x];
  const u32 base = blockIdx.x * 1024u + threadIdx.x * 4u;
#pragma unroll
  for (int i = 0; i < 4; i++)
  {
    const u32 e = base + i;
    if (e < n) { u32* p = v + (MAXFWD ? e : n - 1u - e); *p = op<MAXFWD>(*p, add); }
  }
}

// what the 16 positions of a thread write: visit(k, bytes, firstByteIsCount) in position order
struct Emit
{
  u32 start[16], end[16];    // the segment of each position
};

template<class F>
__device__ __forceinline__ void forEachPosition(const Flags& f, u32 n, u32 i0, u32 headBefore, u32 tailBehind, F&& visit)
{
  // segment starts, front to back; segment ends, back to front
  u32 st[16], en[16];
  u32 s = headBefore;    // (position + 1 of the last head in front of this thread; a head at i0 replaces it at once)
#pragma unroll
  for (int k = 0; k < 16; k++) { if ((f.head >> k) & 1u) s = i0 + (u32)k + 1u; st[k] = s - 1u; }
  u32 e = tailBehind;
#pragma unroll
  for (int k = 15; k >= 0; k--) { if ((f.tail >> k) & 1u) e = i0 + (u32)k; en[k] = e; }
#pragma unroll
  for (int k = 0; k < 16; k++)
  {
    const u32 p = i0 + (u32)k;
    if (p >= n) continue;
    const bool run = (f.run >> (k + 1)) & 1u;
    const bool first = (p - st[k]) % kSeg == 0u;    // the first byte of a token / of a literal chunk
    visit(k, p, run, first, min(kSeg, en[k] - p + 1u));
  }
}

// pass 2: bytes each thread writes
__global__ void __launch_bounds__(256) k_rle_sizes(const u8* __restrict__ b, u32 n, u32 nThreads, const u32* __restrict__ lastHead,
                                                   const u32* __restrict__ firstTail, u32* __restrict__ sizes)
{
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nThreads) return;
  const u32 i0 = 16u * t;
  const Flags f = classify(loadAround(b, n, i0), n, i0);
  u32 bytes = 0;
  forEachPosition(f, n, i0, t ? lastHead[t - 1u] : 0u, t + 1u < nThreads ? firstTail[t + 1u] : kNone,
                  [&](int, u32, bool run, bool first, u32) { bytes += run ? (first ? 3u : 0u) : (first ? 3u : 1u); });
  sizes[t] = bytes;
}

// pass 3: the stream.  Thread 0 also leaves the total (with the end marker) in *sizeOut -- or ~0 if it does not fit `cap`,
// in which case nothing may have been written behind the capacity.
__global__ void __launch_bounds__(256) k_rle_write(const u8* __restrict__ b, u32 n, u32 nThreads, const u32* __restrict__ lastHead,
                                                   const u32* __restrict__ firstTail, const u32* __restrict__ offs, u8* __restrict__ out,
                                                   u32 cap, u32* __restrict__ sizeOut)
{
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nThreads) return;
  const u32 total = offs[nThreads] + 2u;
  if (t == 0)
  {
    *sizeOut = total <= cap ? total : kNone;
    if (total <= cap) { out[total - 2u] = 0x00; out[total - 1u] = 0x80; }    // -32768
  }
  if (total > cap) return;
  const u32 i0 = 16u * t;
  const Around a = loadAround(b, n, i0);
  const Flags f = classify(a, n, i0);
  u32 at = offs[t];
  forEachPosition(f, n, i0, t ? lastHead[t - 1u] : 0u, t + 1u < nThreads ? firstTail[t + 1u] : kNone,
                  [&](int k, u32, bool run, bool first, u32 len)
  {
    const u8 v = (u8)a.at(k);
    if (run)
    {
      if (first) { const u32 c = (u32)(-(int)len) & 0xFFFFu; out[at] = (u8)c; out[at + 1] = (u8)(c >> 8); out[at + 2] = v; at += 3u; }
    }
    else
    {
      if (first) { out[at] = (u8)len; out[at + 1] = (u8)(len >> 8); at += 2u; }
      out[at++] = v;
    }
  });
}

// ================================================================================================
// Decoding (RLE::decompress, RLE.cpp:259-330; the host's rleDecode in codec_common.cpp is the same loop)
// ================================================================================================
// Where a segment starts is known once the segment before it has been read: the same chain as in the block stream, and
// without anything to tell a true segment start from a false one.  So nothing is guessed: every position p of the stream
// gets the step it would make if a segment started there (one look at two bytes), and pointer doubling in LDS turns the steps
// into hops -- first to the end of p's 256-byte sub-piece, then on to the end of its 8 KiB piece --, each with the number of
// mask bytes produced on the way.  One thread then follows the 8 KiB hops from position 0 (a stream of 700 KB: 90 dependent
// loads), one thread per piece the 256-byte hops inside it, and with that every sub-piece on the way knows where the chain
// enters it and which mask byte comes next: a wave walks the sub-piece's twenty-odd segments and writes them, a lane a segment.
constexpr u32 kPiece = 8192u, kSub = 256u;
constexpr u32 kHopEnd = 0xFFFFFFFFu, kHopErr = 0xFFFFFFFEu;    // the stream's end mark was read / the stream is damaged
constexpr u32 kRelEnd = 0xFFFFu, kRelErr = 0xFFFEu;

struct Hop { u32 exit, sum; };    // first segment start at or behind the (sub-)piece's end, or kHopEnd / kHopErr; bytes produced up to there

__global__ void __launch_bounds__(1024) k_rled_hops(const u8* __restrict__ rle, u32 L, Hop* __restrict__ hopSub, Hop* __restrict__ hopPiece)
{
  __shared__ __align__(16) u8 s_b[kPiece + 16];
  __shared__ u16 s_nx[kPiece];    // relative to the piece's start; >= kPiece: outside; kRelEnd / kRelErr
  __shared__ u32 s_sm[kPiece];
  __shared__ u32 s_changed;
  constexpr u32 E = kPiece / 1024u;    // entries per thread
  const u32 B = blockIdx.x * kPiece;
  const u32 nHere = min(kPiece, L - B);
  for (u32 i = threadIdx.x; i < kPiece + 2u; i += 1024u) s_b[i] = (B + i < L) ? rle[B + i] : (u8)0;
  __syncthreads();
  // the single steps
  for (u32 q = 0; q < E; q++)
  {
    const u32 r = q * 1024u + threadIdx.x;
    u32 nx = kRelErr, sm = 0u;
    const u32 p = B + r;
    if (r < nHere && p + 2u <= L)
    {
      const int cnt = (int)(short)((u32)s_b[r] | ((u32)s_b[r + 1] << 8));
      if (cnt == -32768) nx = kRelEnd;
      else
      {
        const u32 n = (u32)(cnt < 0 ? -cnt : cnt), payload = cnt > 0 ? n : 1u;
        if (L - (p + 2u) >= payload + 2u) { nx = r + 2u + payload; sm = n; }    // (+ 2: a count always follows, RLE.cpp:310)
      }
    }
    s_nx[r] = (u16)nx; s_sm[r] = sm;    // (nx <= 8191 + 2 + 32767 < kRelErr)
  }
  __syncthreads();
  // doubling, synchronous rounds: a hop that ends inside the limit is extended by the hop that starts where it ends
  for (int phase = 0; phase < 2; phase++)
  {
    for (;;)
    {
      if (threadIdx.x == 0) s_changed = 0u;
      u32 nx[E], sm[E];
      bool ch = false;
#pragma unroll
      for (u32 q = 0; q < E; q++)
      {
        const u32 r = q * 1024u + threadIdx.x;
        nx[q] = s_nx[r]; sm[q] = s_sm[r];
        const u32 limit = phase == 0 ? min((r | (kSub - 1u)) + 1u, nHere) : nHere;
        if (nx[q] < limit)    // (neither a mark nor outside)
        {
          const u32 j = nx[q];
          nx[q] = s_nx[j]; sm[q] += s_sm[j];
          ch = true;
        }
      }
      __syncthreads();
      if (ch)
      {
        s_changed = 1u;
#pragma unroll
        for (u32 q = 0; q < E; q++) { const u32 r = q * 1024u + threadIdx.x; s_nx[r] = (u16)nx[q]; s_sm[r] = sm[q]; }
      }
      __syncthreads();
      if (!s_changed) break;
      __syncthreads();
    }
    Hop* dst = phase == 0 ? hopSub : hopPiece;
    for (u32 q = 0; q < E; q++)
    {
      const u32 r = q * 1024u + threadIdx.x;
      if (r >= nHere) continue;
      const u32 nx = s_nx[r];
      Hop h;
      h.exit = nx == kRelEnd ? kHopEnd : nx == kRelErr ? kHopErr : B + nx;
      h.sum = s_sm[r];
      dst[B + r] = h;
    }
    __syncthreads();
  }
}

// entry of the chain into every piece it touches: (position, mask bytes produced before); untouched pieces keep kNone
__global__ void __launch_bounds__(64) k_rled_chain(const Hop* __restrict__ hopPiece, u32 L, u32 outBytes, uint2* __restrict__ pieceEntry,
                                                   DeviceStatus* st)
{
  if (threadIdx.x != 0) return;
  u32 pos = 0;
  u64 out = 0;
  for (;;)
  {
    if (pos >= L) { raiseError(st, kFailed, 0x52000001u); return; }
    pieceEntry[pos / kPiece] = make_uint2(pos, (u32)out);
    const Hop h = hopPiece[pos];
    out += h.sum;
    if (out > outBytes || h.exit == kHopErr) { raiseError(st, kFailed, 0x52000002u); return; }
    if (h.exit == kHopEnd) return;
    pos = h.exit;
  }
}

__global__ void __launch_bounds__(64) k_rled_sub_entries(const Hop* __restrict__ hopSub, u32 L, const uint2* __restrict__ pieceEntry,
                                                         uint2* __restrict__ subEntry, const DeviceStatus* st)
{
  const u32 b = blockIdx.x * 64u + threadIdx.x;
  if (b >= (L + kPiece - 1u) / kPiece || st->error) return;
  const uint2 e = pieceEntry[b];
  if (e.x == kNone) return;
  u32 pos = e.x, out = e.y;
  const u32 end = min((b + 1u) * kPiece, L);
  while (pos < end)
  {
    subEntry[pos / kSub] = make_uint2(pos, out);
    const Hop h = hopSub[pos];
    out += h.sum;
    if (h.exit >= kHopErr) break;
    pos = h.exit;
  }
}

// a wave per sub-piece the chain enters
__global__ void __launch_bounds__(64) k_rled_expand(const u8* __restrict__ rle, u32 L, const uint2* __restrict__ subEntry,
                                                    u8* __restrict__ bits, u32 outBytes, const DeviceStatus* st)
{
  constexpr u32 kMaxSeg = kSub / 3u + 2u;    // (a segment is three bytes at least; one whose count is 0 too)
  __shared__ u8 s_b[kSub + 8];
  __shared__ u32 s_pos[kMaxSeg], s_out[kMaxSeg];
  __shared__ int s_cnt[kMaxSeg];
  __shared__ u32 s_n;
  const u32 s = blockIdx.x, lane = threadIdx.x;
  const uint2 e = subEntry[s];
  if (e.x == kNone || st->error) return;
  const u32 S = s * kSub, end = min(S + kSub, L);
  for (u32 i = lane; i < kSub + 2u; i += 64u) s_b[i] = (S + i < L) ? rle[S + i] : (u8)0;
  __syncthreads();
  if (lane == 0)
  {
    u32 pos = e.x, out = e.y, n = 0;
    while (pos < end && pos + 2u <= L && n < kMaxSeg)
    {
      const int cnt = (int)(short)((u32)s_b[pos - S] | ((u32)s_b[pos - S + 1] << 8));
      if (cnt == -32768) break;
      const u32 len = (u32)(cnt < 0 ? -cnt : cnt), payload = cnt > 0 ? len : 1u;
      if (L - (pos + 2u) < payload + 2u) break;    // (k_rled_chain has raised the error)
      s_pos[n] = pos + 2u; s_out[n] = out; s_cnt[n] = cnt; n++;
      out += len; pos += 2u + payload;
    }
    s_n = n;
  }
  __syncthreads();
  const u32 n = s_n;
  constexpr u32 kBig = 256u;
  // short segments: a lane each
  for (u32 k = lane; k < n; k += 64u)
  {
    const int cnt = s_cnt[k];
    const u32 len = (u32)(cnt < 0 ? -cnt : cnt);
    const u32 o = s_out[k];
    if (len == 0u || len >= kBig || (u64)o + len > outBytes) continue;
    const u8* src = rle + s_pos[k];
    u8* dst = bits + o;
    if (cnt < 0)
    {
      const u8 v = src[0];
      const u64 v8 = 0x0101010101010101ull * v;
      u32 i = 0;
      for (; i < len && (((uintptr_t)(dst + i)) & 7u); i++) dst[i] = v;
      for (; i + 8u <= len; i += 8u) *reinterpret_cast<u64*>(dst + i) = v8;
      for (; i < len; i++) dst[i] = v;
    }
    else for (u32 i = 0; i < len; i++) dst[i] = src[i];
  }
  // long ones: the wave together, 8 bytes a lane where the destination allows
  for (u32 k = 0; k < n; k++)
  {
    const int cnt = s_cnt[k];
    const u32 len = (u32)(cnt < 0 ? -cnt : cnt);
    const u32 o = s_out[k];
    if (len < kBig || (u64)o + len > outBytes) continue;
    const u8* src = rle + s_pos[k];
    u8* dst = bits + o;
    const u32 head = min(len, (u32)((8u - ((uintptr_t)dst & 7u)) & 7u));
    const u32 words = (len - head) / 8u, tail = len - head - 8u * words;
    if (cnt < 0)
    {
      const u8 v = src[0];
      const u64 v8 = 0x0101010101010101ull * v;
      if (lane < head) dst[lane] = v;
      for (u32 w = lane; w < words; w += 64u) *reinterpret_cast<u64*>(dst + head + 8u * w) = v8;
      if (lane < tail) dst[head + 8u * words + lane] = v;
    }
    else
    {
      if (lane < head) dst[lane] = src[lane];
      for (u32 w = lane; w < words; w += 64u)
      {
        u64 x = 0;
#pragma unroll
        for (u32 q = 0; q < 8u; q++) x |= (u64)src[head + 8u * w + q] << (8u * q);
        *reinterpret_cast<u64*>(dst + head + 8u * w) = x;
      }
      if (lane < tail) dst[head + 8u * words + lane] = src[head + 8u * words + lane];
    }
  }
}

}    // namespace

size_t maskRleScratchBytes(size_t nBytes)
{
  const size_t nThreads = (nBytes + 15) / 16;
  return (4 * (nThreads + 8) + 3 * ((nThreads + 1023) / 1024 + 8) + 1024) * sizeof(u32);
}

void launchMaskRle(const u8* bits, u32 nBytes, u8* out, u32 cap, u32* sizeOut, u8* scratch, hipStream_t st)
{
  const u32 nThreads = (nBytes + 15u) / 16u, nWG = (nThreads + 255u) / 256u, nPart = (nThreads + 1023u) / 1024u;
  u32* lastHead = reinterpret_cast<u32*>(scratch);
  u32* firstTail = lastHead + nThreads + 8;
  u32* sizes = firstTail + nThreads + 8;
  u32* offs = sizes + nThreads + 8;
  u32* part = offs + nThreads + 8;    // [3][nPart + 8] + the sum scan's scratch
  hipLaunchKernelGGL(k_rle_flags, dim3(nWG), dim3(256), 0, st, bits, nBytes, nThreads, lastHead, firstTail);
  hipLaunchKernelGGL(k_rle_scan_local<true>, dim3(nPart), dim3(256), 0, st, lastHead, nThreads, part);
  hipLaunchKernelGGL(k_rle_scan_local<false>, dim3(nPart), dim3(256), 0, st, firstTail, nThreads, part + nPart + 8);
  if (nPart > 1)
  {
    hipLaunchKernelGGL(k_rle_scan_partials<true>, dim3(1), dim3(256), 0, st, part, nPart);
    hipLaunchKernelGGL(k_rle_scan_partials<false>, dim3(1), dim3(256), 0, st, part + nPart + 8, nPart);
    hipLaunchKernelGGL(k_rle_scan_add<true>, dim3(nPart), dim3(256), 0, st, lastHead, nThreads, (const u32*)part);
    hipLaunchKernelGGL(k_rle_scan_add<false>, dim3(nPart), dim3(256), 0, st, firstTail, nThreads, (const u32*)(part + nPart + 8));
  }
  hipLaunchKernelGGL(k_rle_sizes, dim3(nWG), dim3(256), 0, st, bits, nBytes, nThreads, (const u32*)lastHead, (const u32*)firstTail, sizes);
  launchExclusiveScan(sizes, offs, nThreads, part + 2 * (nPart + 8), st);
  hipLaunchKernelGGL(k_rle_write, dim3(nWG), dim3(256), 0, st, bits, nBytes, nThreads, (const u32*)lastHead, (const u32*)firstTail,
                     (const u32*)offs, out, cap, sizeOut);
}

size_t maskRleDecodeScratchBytes(size_t rleBytes)
{
  const size_t nPiece = (rleBytes + kPiece - 1) / kPiece, nSub = (rleBytes + kSub - 1) / kSub;
  return 2 * (rleBytes + 8) * sizeof(Hop) + (nPiece + nSub + 16) * sizeof(uint2) + 256;
}

// bits[0 .. outBytes) = the decoded stream, zeros behind what it holds; a damaged stream raises kFailed in *status
void launchMaskRleDecode(const u8* rle, u32 rleBytes, u8* bits, u32 outBytes, u8* scratch, DeviceStatus* status, hipStream_t st)
{
  hipMemsetAsync(bits, 0, outBytes, st);
  if (rleBytes < 2u) return;
  const u32 nPiece = (rleBytes + kPiece - 1u) / kPiece, nSub = (rleBytes + kSub - 1u) / kSub;
  Hop* hopSub = reinterpret_cast<Hop*>(((uintptr_t)scratch + 15u) & ~(uintptr_t)15u);
  Hop* hopPiece = hopSub + rleBytes + 8;
  uint2* pieceEntry = reinterpret_cast<uint2*>(hopPiece + rleBytes + 8);
  uint2* subEntry = pieceEntry + nPiece + 8;
  hipMemsetAsync(pieceEntry, 0xFF, ((size_t)nPiece + 8 + nSub + 8) * sizeof(uint2), st);
  hipLaunchKernelGGL(k_rled_hops, dim3(nPiece), dim3(1024), 0, st, rle, rleBytes, hopSub, hopPiece);
  hipLaunchKernelGGL(k_rled_chain, dim3(1), dim3(64), 0, st, (const Hop*)hopPiece, rleBytes, outBytes, pieceEntry, status);
  hipLaunchKernelGGL(k_rled_sub_entries, dim3((nPiece + 63u) / 64u), dim3(64), 0, st, (const Hop*)hopSub, rleBytes, (const uint2*)pieceEntry,
                     subEntry, (const DeviceStatus*)status);
  hipLaunchKernelGGL(k_rled_expand, dim3(nSub), dim3(64), 0, st, rle, rleBytes, (const uint2*)subEntry, bits, outBytes, (const DeviceStatus*)status);
}

}    // namespace lerc
