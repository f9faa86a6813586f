// gather_rccl.cpp -- the mosaic job's ONE exchange step below Python: the gather of the ranks' compressed blobs on the rank that writes
// the container, over RCCL (xGMI inside a node).  SURVEY.md section 8(e) is the contract; the reference has nothing here (a LERC blob
// is a byte string, whoever tiles a mosaic moves the strings himself).
//
//   every rank:  ncclAllGather of ONE 64-bit word, the length of its message (its arena as lerc_amd_encode_tiles_device left it, the
//                per-tile table in front if the caller put it there: one message a rank)
//   non-root:    ncclSend of the message, posted at once -- a sender needs nobody's length but its own
//   root:        reads the lengths (the only host wait, on the root only, 8 bytes a rank through pinned memory), places the messages
//                back to back at 16-byte aligned offsets, posts one group of ncclRecv, copies its own message on the same stream
// All of it is enqueued on the caller's stream; the function returns when the transfers are POSTED (the caller decodes its own tiles
// meanwhile and waits for the stream when it wants the bytes).
//
// librccl is opened lazily with dlopen (single-GPU users never load it; inside a PyTorch process the loader hands back the RCCL that
// torch.distributed already uses, by its soname).
#include "codec.h"
#include "../../include/lerc_amd_device.h"

#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace lerc {

namespace {

typedef int ncclResult;                       // ncclSuccess == 0
enum { kNcclUint8 = 1, kNcclUint64 = 5 };     // ncclDataType_t (rccl.h:455-470)
struct Rccl
{
  void* lib = nullptr;
  ncclResult (*allGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  ncclResult (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  ncclResult (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  ncclResult (*groupStart)() = nullptr;
  ncclResult (*groupEnd)() = nullptr;
  ncclResult (*commCount)(void*, int*) = nullptr;
  ncclResult (*commUserRank)(void*, int*) = nullptr;
  const char* (*errorString)(ncclResult) = nullptr;
  std::string why;
};

Rccl& rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []()
  {
    const char* env = getenv("LERC_AMD_RCCL");
    const char* names[] = { env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names)
    {
      if (!n || !*n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      r.why = dlerror();
    }
    if (!r.lib) return;
    auto sym = [&](const char* name) -> void* { void* p = dlsym(r.lib, name); if (!p) r.why = std::string("librccl lacks ") + name; return p; };
    r.allGather = (decltype(r.allGather))sym("ncclAllGather");
    r.send = (decltype(r.send))sym("ncclSend");
    r.recv = (decltype(r.recv))sym("ncclRecv");
    r.groupStart = (decltype(r.groupStart))sym("ncclGroupStart");
    r.groupEnd = (decltype(r.groupEnd))sym("ncclGroupEnd");
    r.commCount = (decltype(r.commCount))sym("ncclCommCount");
    r.commUserRank = (decltype(r.commUserRank))sym("ncclCommUserRank");
    r.errorString = (decltype(r.errorString))sym("ncclGetErrorString");
    if (!r.allGather || !r.send || !r.recv || !r.groupStart || !r.groupEnd || !r.commCount || !r.commUserRank) { dlclose(r.lib); r.lib = nullptr; }
  });
  return r;
}

// the lengths' scratch: nRanks words on the device and their pinned mirror, kept per host thread
struct LengthScratch
{
  u64* d = nullptr; u64* h = nullptr; size_t cap = 0;
  hipEvent_t lengthsHome = nullptr;    // recorded behind the lengths' copy to the host: what the host waits for is that, not the transfers posted behind it
  ~LengthScratch() { if (d) hipFree(d); if (h) hipHostFree(h); if (lengthsHome) hipEventDestroy(lengthsHome); }
  bool need(size_t n)
  {
    if (n <= cap) return true;
    if (d) hipFree(d);
    if (h) hipHostFree(h);
    d = nullptr; h = nullptr; cap = 0;
    if (hipMalloc((void**)&d, (n + 8) * 8) != hipSuccess) return false;
    if (hipHostMalloc((void**)&h, (n + 8) * 8, hipHostMallocDefault) != hipSuccess) { hipFree(d); d = nullptr; return false; }
    cap = n + 8;
    return true;
  }
};

}    // namespace

u32 gatherBlobsRccl(void* comm, int root, const void* dMessage, u64 nBytes, void* dRootBuffer, u64 rootCapacity, u64* hLengths, u64* hOffsets,
                    hipStream_t st, std::string& err)
{
  Rccl& R = rccl();
  if (!R.lib) { err = "lerc_amd_gather_blobs: librccl could not be opened (" + R.why + ")"; return kFailed; }
  int nRanks = 0, rank = -1;
  if (R.commCount(comm, &nRanks) != 0 || R.commUserRank(comm, &rank) != 0 || nRanks < 1 || rank < 0 || rank >= nRanks)
  { err = "lerc_amd_gather_blobs: not a communicator"; return kWrongParam; }
  if (root < 0 || root >= nRanks || (nBytes != 0 && !dMessage) || (rank == root && !dRootBuffer)) { err = "lerc_amd_gather_blobs: wrong parameter"; return kWrongParam; }
  auto fail = [&](const char* what, ncclResult rc) -> u32
  {
    char msg[160];
    snprintf(msg, sizeof(msg), "lerc_amd_gather_blobs: %s failed (%s)", what, R.errorString ? R.errorString(rc) : "?");
    err = msg;
    return kFailed;
  };
  static thread_local LengthScratch sc;
  if (!sc.need((size_t)nRanks + 1)) { err = "lerc_amd_gather_blobs: no memory for the lengths"; return kFailed; }

  // ---- the lengths: this rank's word travels through pinned memory onto the stream, then one all-gather (in place)
  sc.h[nRanks] = nBytes;
  if (hipMemcpyAsync(sc.d + rank, sc.h + nRanks, 8, hipMemcpyHostToDevice, st) != hipSuccess) { err = "lerc_amd_gather_blobs: copy failed"; return kFailed; }
  ncclResult rc = R.allGather(sc.d + rank, sc.d, 1, kNcclUint64, comm, st);
  if (rc != 0) return fail("ncclAllGather", rc);
  if (hipMemcpyAsync(sc.h, sc.d, (size_t)nRanks * 8, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "lerc_amd_gather_blobs: copy failed"; return kFailed; }
  if (!sc.lengthsHome && hipEventCreateWithFlags(&sc.lengthsHome, hipEventDisableTiming) != hipSuccess) { sc.lengthsHome = nullptr; err = "lerc_amd_gather_blobs: no event"; return kFailed; }
  if (hipEventRecord(sc.lengthsHome, st) != hipSuccess) { err = "lerc_amd_gather_blobs: the stream failed"; return kFailed; }

  // ---- a sender posts its message at once; everybody learns the lengths (the root needs them to post its receives, the others
  // hand them to the caller -- their wait lies BEHIND their send's posting)
  if (rank != root && nBytes != 0)
  {
    rc = R.send(dMessage, (size_t)nBytes, kNcclUint8, root, comm, st);
    if (rc != 0) return fail("ncclSend", rc);
  }
  if (hipEventSynchronize(sc.lengthsHome) != hipSuccess) { err = "lerc_amd_gather_blobs: the stream failed"; return kFailed; }    // (NOT the stream: a sender's message is on it)
  u64 at = 0, ownAt = 0;
  for (int r = 0; r < nRanks; r++)
  {
    if (hLengths) hLengths[r] = sc.h[r];
    if (hOffsets) hOffsets[r] = at;
    if (r == root) ownAt = at;
    at += (sc.h[r] + 15ull) & ~15ull;
  }
  if (hOffsets) hOffsets[nRanks] = at;
  if (rank != root) return kOk;
  if (at > rootCapacity) { err = "lerc_amd_gather_blobs: the root's buffer is too small"; return kBufferTooSmall; }    // (the senders' messages stay posted: the job is over)

  // ---- the root: its own message by a copy, the others' by one group of receives
  if (nBytes != 0 && hipMemcpyAsync((u8*)dRootBuffer + ownAt, dMessage, (size_t)nBytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
  { err = "lerc_amd_gather_blobs: copy failed"; return kFailed; }
  if (nRanks == 1) return kOk;
  rc = R.groupStart();
  if (rc != 0) return fail("ncclGroupStart", rc);
  at = 0;
  for (int r = 0; r < nRanks; r++)
  {
    if (r != root && sc.h[r] != 0)
    {
      rc = R.recv((u8*)dRootBuffer + at, (size_t)sc.h[r], kNcclUint8, r, comm, st);
      if (rc != 0) { R.groupEnd(); return fail("ncclRecv", rc); }
    }
    at += (sc.h[r] + 15ull) & ~15ull;
  }
  rc = R.groupEnd();
  if (rc != 0) return fail("ncclGroupEnd", rc);
  return kOk;
}

}    // namespace lerc
