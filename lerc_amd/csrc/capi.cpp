// capi.cpp -- extern "C" boundary of liblerc_amd.so (include/lerc_amd.h).
// Argument validation and status codes follow the reference's Lerc_c_api_impl.cpp:33-304; host
// buffers are staged through HBM, all codec work happens in codec_encode.cpp / codec_decode.cpp.
#include "../../include/lerc_amd.h"
#include "../../include/lerc_amd_device.h"
#include "codec.h"

#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <deque>
#include <new>
#include <vector>

using namespace lerc;

// an operation of the asynchronous device API that has been enqueued but not waited for
struct AsyncOp
{
  unsigned ticket = 0;
  bool isEncode = false;
  EncodeRequest er;
  DecodeRequest dr;
  bool streamed = false;    // its streaming kernels and the copy of their verdict are on the stream
  int form = 0;             // a decode: which streaming form (DecodeRequest::maxForm)
  u32 epoch = 0;
  bool done = false;
  u32 status = kOk, bytes = 0;
};

// A decode that ends in Failed leaves zeros in the caller's device buffers: the streaming kernels write pixels while the blob's
// checksum is still being summed (the reference checks first, Lerc2.cpp:592-601, and leaves the buffer alone).
static void wipeDecodeOutputs(Context& ctx, const DecodeRequest& rq)
{
  hipStream_t st = ctx.activeStream();
  const size_t nPix = (size_t)rq.nRows * rq.nCols;
  if (rq.dOut) hipMemsetAsync(rq.dOut, 0, nPix * rq.nDepth * rq.nBands * (size_t)dtSize(rq.dt), st);
  if (rq.dValidBytes && rq.nMasks > 0) hipMemsetAsync(rq.dValidBytes, 0, nPix * (size_t)rq.nMasks, st);
  hipStreamSynchronize(st);
}

struct lerc_amd_context
{
  Context ctx;
  std::deque<AsyncOp> ops;  // in enqueue order; results stay until lerc_amd_finish has handed them out
  unsigned nextTicket = 1;
  // staging buffers for the host-pointer API (grown on demand, owned by the context)
  void* io[3] = { nullptr, nullptr, nullptr };
  size_t ioCap[3] = { 0, 0, 0 };
  ~lerc_amd_context() { for (void* p : io) if (p) hipFree(p); }
  void* stage(int slot, size_t bytes)
  {
    if (bytes <= ioCap[slot]) return io[slot];
    if (io[slot]) { hipStreamSynchronize(ctx.activeStream()); hipFree(io[slot]); io[slot] = nullptr; ioCap[slot] = 0; }
    const size_t want = bytes + bytes / 16 + 4096;
    if (hipMalloc(&io[slot], want) != hipSuccess) return nullptr;
    ioCap[slot] = want;
    return io[slot];
  }
};

namespace {

// The stock entry points are re-entrant across host threads like the reference's (Lerc.cpp:448,640: no global state):
// every thread that calls them owns one context -- workspace slab, three staging buffers, pinned mirror, events --
// and gives it back when the thread ends.  The main thread's holder is destroyed while the process exits, when the HIP
// runtime may be half gone already (hipFree / hipStreamDestroy have been seen to hang there on some ROCm versions): its
// context is left to the operating system, like everything else the process owns.
struct ThreadHolder
{
  lerc_amd_context* h = nullptr;
  ~ThreadHolder()
  {
    const bool mainThread = (long)syscall(SYS_gettid) == (long)getpid();
    if (h && !mainThread)
    {
      hipStreamSynchronize(h->ctx.activeStream());    // (operations still queued: their buffers go away with the context)
      delete h;
    }
    h = nullptr;
  }
};

lerc_amd_context* threadHandle()
{
  static thread_local ThreadHolder holder;
  if (!holder.h) holder.h = new (std::nothrow) lerc_amd_context();
  return (holder.h && holder.h->ctx.ok()) ? holder.h : nullptr;
}

bool masksArgOk(int nMasks, int nBands, const void* pValidBytes)
{
  return (nMasks == 0 || nMasks == 1 || nMasks == nBands) && !(nMasks > 0 && !pValidBytes);
}

bool dimsOk(int nDepth, int nCols, int nRows, size_t elemSize)    // Lerc.cpp:1622-1639
{
  if (nDepth <= 0 || nCols <= 0 || nRows <= 0) return false;
  const u64 nPix = (u64)nRows * nCols, lim = INT_MAX, bpp = elemSize;
  return !(nPix > lim || bpp > lim || bpp * nDepth > lim || bpp * nDepth * nPix > lim);
}

bool usesNoData(const unsigned char* pUsesNoData, int nBands)
{
  if (!pUsesNoData) return false;
  for (int i = 0; i < nBands; i++) if (pUsesNoData[i]) return true;
  return false;
}

// shared by lerc_encode / lerc_computeCompressedSize: host pointers in, blob (optionally) out
lerc_status encodeHost(const void* pData, unsigned dataType, int nDepth, int nCols, int nRows, int nBands, int nMasks,
                       const unsigned char* pValidBytes, double maxZErr, unsigned char* pOut, unsigned outSize,
                       unsigned* result, bool sizeOnly, const unsigned char* pUsesNoData = nullptr, const double* noDataValues = nullptr,
                       int version = kCodecVersion)
{
  lerc_amd_context* h = threadHandle();
  if (!h) return kFailed;
  Context& ctx = h->ctx;
  hipStream_t st = ctx.activeStream();
  const size_t tb = (size_t)dtSize((int)dataType);
  if (!dimsOk(nDepth, nCols, nRows, tb)) return kDimsTooLarge;
  const size_t nPix = (size_t)nRows * nCols;
  const size_t dataBytes = nPix * nDepth * tb * nBands, maskBytes = (size_t)nMasks * nPix;
  u8* dData = (u8*)h->stage(0, dataBytes);
  u8* dMask = nMasks ? (u8*)h->stage(1, maskBytes) : nullptr;
  if (!dData || (nMasks && !dMask)) return kFailed;
  hipMemcpyAsync(dData, pData, dataBytes, hipMemcpyHostToDevice, st);
  if (nMasks) hipMemcpyAsync(dMask, pValidBytes, maskBytes, hipMemcpyHostToDevice, st);

  EncodeRequest rq;
  rq.dData = dData; rq.dValidBytes = dMask; rq.dt = (int)dataType; rq.nDepth = nDepth; rq.nCols = nCols; rq.nRows = nRows;
  rq.nBands = nBands; rq.nMasks = nMasks; rq.maxZErr = maxZErr;
  rq.hUsesNoData = pUsesNoData; rq.hNoDataValues = noDataValues;
  rq.version = version;
  u32 needed = 0, written = 0;
  if (sizeOnly)
  {
    const u32 rc = encodeDevice(ctx, rq, needed, written);
    if (rc == kOk) *result = needed;
    return rc;
  }
  // a blob never exceeds raw pixels + mask RLE + header + the per-depth ranges of codec >= 4 (2 * nDepth values) + a few
  // hundred bytes per band (one-sweep fallback, Lerc2.cpp:364)
  const u64 bound = (u64)nBands * (nPix * nDepth * tb + nPix / 4 + 2 * (u64)nDepth * tb + 4096);
  const u32 cap = (u32)std::min<u64>(outSize, bound);
  u8* dOut = (u8*)h->stage(2, cap);
  if (!dOut) return kFailed;
  rq.dOut = dOut; rq.outCapacity = cap;
  memset(pOut, 0, outSize);    // Lerc.cpp:374
  // Small single-band rasters the streaming kernels take: the blob's way back is enqueued behind the kernels before its
  // size is known -- the whole (small) output buffer into pinned memory -- and this thread waits once instead of twice;
  // the bytes that count are then copied on the host.  (Two waits and the gap between them are a third of such a call.)
  if (nBands == 1 && nMasks == 0 && cap <= (2u << 20))
  {
    u8* slot = (u8*)ctx.pinned(Context::kAsyncSlotBytes);
    u8* hStage = (u8*)ctx.pinnedAux(cap);
    if (slot && hStage && encodeEnqueueStreaming(ctx, rq, slot))
    {
      if (hipMemcpyAsync(hStage, dOut, cap, hipMemcpyDeviceToHost, st) != hipSuccess || !ctx.sync()) return kFailed;
      bool redo = false;
      u32 status = kOk;
      encodeStreamingVerdict(ctx, rq, slot, redo, status, needed, written);
      if (!redo)
      {
        if (status != kOk) return (lerc_status)status;
        memcpy(pOut, hStage, written);
        *result = written;
        return kOk;
      }
    }
  }
  const u32 rc = encodeDevice(ctx, rq, needed, written);
  if (rc != kOk) return rc;
  if (hipMemcpyAsync(pOut, dOut, written, hipMemcpyDeviceToHost, st) != hipSuccess) return kFailed;
  if (hipStreamSynchronize(st) != hipSuccess) return kFailed;    // (not the polling Context::sync(): the copy into the caller's pageable
                                                                 // buffer is driven by this wait)
  *result = written;
  return kOk;
}

lerc_status decodeHost(const unsigned char* blob, unsigned blobSize, int nMasks, unsigned char* pValidBytes, int nDepth,
                       int nCols, int nRows, int nBands, unsigned dataType, void* pData, bool toDouble,
                       unsigned char* pUsesNoData = nullptr, double* noDataValues = nullptr)
{
  lerc_amd_context* h = threadHandle();
  if (!h) return kFailed;
  Context& ctx = h->ctx;
  hipStream_t st = ctx.activeStream();
  const size_t tb = (size_t)dtSize((int)dataType);
  if (!dimsOk(nDepth, nCols, nRows, tb)) return kDimsTooLarge;
  const size_t nPix = (size_t)nRows * nCols, nVals = nPix * nDepth * nBands;
  const size_t outBytes = nVals * tb, maskBytes = (size_t)nMasks * nPix;
  const bool widen = toDouble && dataType != DT_Double;
  u8* dOut = (u8*)h->stage(0, outBytes + (widen ? nVals * 8 + 256 : 0));
  u8* dMask = nMasks ? (u8*)h->stage(1, maskBytes) : nullptr;
  if (!dOut || (nMasks && !dMask)) return kFailed;

  if (isLerc1(blob, blobSize))
  {
    // Lerc1 leaves pixels that are not valid as the caller's buffer had them (Lerc::Convert, Lerc.cpp:795-845): the staging
    // buffer starts out as a copy of it (lerc_decodeToDouble decodes into the tail of the double buffer, Lerc_c_api_impl.cpp:288-300)
    const u8* src = (const u8*)pData + (widen ? nVals * (8 - tb) : 0);
    hipMemcpyAsync(dOut, src, outBytes, hipMemcpyHostToDevice, st);
  }
  DecodeRequest rq;
  rq.hBlob = blob; rq.blobSize = blobSize; rq.dt = (int)dataType; rq.nDepth = nDepth; rq.nCols = nCols; rq.nRows = nRows;
  rq.nBands = nBands; rq.nMasks = nMasks; rq.dOut = dOut; rq.dValidBytes = dMask;
  rq.hUsesNoData = pUsesNoData; rq.hNoDataValues = noDataValues;
  bool triedOne = false;
  // What the streaming kernels can take -- the header says so: one band, every pixel valid, 8 x 8 blocks, nDepth 1 -- goes
  // up, through the kernels and back in one go: upload, kernels, verdict and the pixels' way back are enqueued together
  // and this thread waits once (two waits and the gap between them are a quarter of a small raster's call).
  {
    Header hd;
    size_t used = 0;
    const size_t nPixHd = (size_t)nRows * nCols;
    if (!widen && nBands == 1 && nDepth == 1 && !pUsesNoData && readHeader(blob, blobSize, hd, used) && hd.version >= 3 && hd.nRows == nRows
      && hd.nCols == nCols && hd.nDepth == 1 && hd.dt == (int)dataType && hd.mbSize == 8 && hd.nBlobsMore == 0
      && (size_t)hd.numValid == nPixHd && (unsigned)hd.blobSize <= blobSize && hd.maxZErr > 0 && hd.zMin != hd.zMax)
    {
      u8* dBlob = (u8*)h->stage(2, (size_t)blobSize + 256);
      if (dBlob && hipMemcpyAsync(dBlob, blob, blobSize, hipMemcpyHostToDevice, st) == hipSuccess)
      {
        DecodeRequest rs = rq;
        rs.hBlob = nullptr; rs.dBlob = dBlob;
        bool handled = false, tried = false;
        const u32 src = decodeSpeculativeToHost(ctx, rs, pData, outBytes, nMasks ? pValidBytes : nullptr, maskBytes, handled, tried);
        if (src != kOk) return src;
        if (handled) return kOk;
        triedOne = tried;    // (nothing enqueued: no form has refused the blob, nothing was written)
      }
    }
  }
  if (triedOne) rq.maxForm = ctx.lastStreamForm - 1;    // (that streaming form has just refused this blob)
  if (rq.maxForm <= 0) rq.noStreaming = true;
  const u32 rc = decodeDevice(ctx, rq);
  if (rc != kOk)
  {
    // (the reference checks the checksum before it writes a pixel, Lerc2.cpp:592-601; here the streaming kernels' pixels were on
    // their way into the caller's buffers while the checksum was being summed: what did not pass is wiped -- a failed decode
    // leaves zeros, never pixels of a blob that is damaged)
    if (triedOne)
    {
      memset(pData, 0, toDouble ? nVals * sizeof(double) : outBytes);
      if (nMasks && pValidBytes) memset(pValidBytes, 0, maskBytes);
    }
    return rc;
  }
  if (widen)
  {
    double* dWide = (double*)(dOut + ((outBytes + 255) / 256) * 256);
    launchWidenToDouble((int)dataType, dOut, dWide, (i64)nVals, st);
    hipMemcpyAsync(pData, dWide, nVals * 8, hipMemcpyDeviceToHost, st);
  }
  else hipMemcpyAsync(pData, dOut, outBytes, hipMemcpyDeviceToHost, st);
  if (nMasks) hipMemcpyAsync(pValidBytes, dMask, maskBytes, hipMemcpyDeviceToHost, st);
  return hipStreamSynchronize(st) == hipSuccess ? (lerc_status)kOk : (lerc_status)kFailed;
}

}    // namespace

extern "C" {

lerc_status lerc_computeCompressedSize_4D(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
  int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes,
  const unsigned char* pUsesNoData, const double* noDataValues)
{
  if (!numBytes) return kWrongParam;
  *numBytes = 0;
  if (!pData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0) return kWrongParam;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return kWrongParam;
  if (usesNoData(pUsesNoData, nBands) && !noDataValues) return kWrongParam;
  return encodeHost(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, nullptr, 0, numBytes, true, pUsesNoData,
                    noDataValues);
}

lerc_status lerc_encode_4D(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows, int nBands,
  int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer, unsigned int outBufferSize,
  unsigned int* nBytesWritten, const unsigned char* pUsesNoData, const double* noDataValues)
{
  if (!nBytesWritten) return kWrongParam;
  *nBytesWritten = 0;
  if (!pData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0
    || !pOutBuffer || !outBufferSize)
    return kWrongParam;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return kWrongParam;
  if (usesNoData(pUsesNoData, nBands) && !noDataValues) return kWrongParam;
  return encodeHost(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, pOutBuffer, outBufferSize,
                    nBytesWritten, false, pUsesNoData, noDataValues);
}

lerc_status lerc_computeCompressedSizeForVersion(const void* pData, int codecVersion, unsigned int dataType, int nDepth,
  int nCols, int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes)
{
  if (!numBytes) return kWrongParam;
  *numBytes = 0;
  if (codecVersion > kCodecVersion) return kWrongParam;
  if (codecVersion >= 0 && codecVersion < kCodecVersion)    // Lerc.cpp:339-347 -> EncodeInternal_v5
  {
    if (!pData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0) return kWrongParam;
    if (!masksArgOk(nMasks, nBands, pValidBytes)) return kWrongParam;
    if (codecVersion < 2) return kWrongParam;    // Lerc2::SetEncoderToOldVersion (Lerc2.cpp:52-62)
    return encodeHost(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, nullptr, 0, numBytes, true, nullptr,
                      nullptr, codecVersion);
  }
  return lerc_computeCompressedSize_4D(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, numBytes,
                                       nullptr, nullptr);
}

lerc_status lerc_encodeForVersion(const void* pData, int codecVersion, unsigned int dataType, int nDepth, int nCols,
  int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer,
  unsigned int outBufferSize, unsigned int* nBytesWritten)
{
  if (!nBytesWritten) return kWrongParam;
  *nBytesWritten = 0;
  if (codecVersion > kCodecVersion) return kWrongParam;
  if (codecVersion >= 0 && codecVersion < kCodecVersion)    // Lerc.cpp:378-386
  {
    if (!pData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0
      || !pOutBuffer || !outBufferSize)
      return kWrongParam;
    if (!masksArgOk(nMasks, nBands, pValidBytes)) return kWrongParam;
    if (codecVersion < 2) return kWrongParam;    // Lerc2::SetEncoderToOldVersion (Lerc2.cpp:52-62)
    return encodeHost(pData, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, pOutBuffer, outBufferSize,
                      nBytesWritten, false, nullptr, nullptr, codecVersion);
  }
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
  return lerc_encodeForVersion(pData, -1, dataType, nDepth, nCols, nRows, nBands, nMasks, pValidBytes, maxZErr, pOutBuffer,
                               outBufferSize, nBytesWritten);
}

lerc_status lerc_getBlobInfo(const unsigned char* pLercBlob, unsigned int blobSize, unsigned int* infoArray,
  double* dataRangeArray, int infoArraySize, int dataRangeArraySize)
{
  if (!pLercBlob || !blobSize || (!infoArray && !dataRangeArray) || ((infoArraySize <= 0) && (dataRangeArraySize <= 0)))
    return kWrongParam;
  BlobInfo li;
  u32 e = getBlobInfo(pLercBlob, blobSize, li);
  if (e != kOk && isLerc1(pLercBlob, blobSize))    // legacy Lerc1: the bands have to be decoded to know their ranges (Lerc.cpp:184-266)
  {
    lerc_amd_context* h = threadHandle();
    if (!h) return kFailed;
    e = lerc1BlobInfo(h->ctx, pLercBlob, blobSize, li, nullptr, nullptr, 0);
  }
  if (e != kOk) return e;
  if (infoArray)
  {
    const unsigned v[11] = { (unsigned)li.version, (unsigned)li.dt, (unsigned)li.nDepth, (unsigned)li.nCols, (unsigned)li.nRows,
      (unsigned)li.nBands, (unsigned)li.numValid, li.blobSize, (unsigned)li.nMasks, (unsigned)li.nDepth, (unsigned)li.nUsesNoData };
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
  return kOk;
}

lerc_status lerc_getDataRanges(const unsigned char* pLercBlob, unsigned int blobSize, int nDepth, int nBands,
  double* pMins, double* pMaxs)
{
  if (!pLercBlob || !blobSize || !pMins || !pMaxs || nDepth <= 0 || nBands <= 0) return kWrongParam;
  BlobInfo li;
  const u32 e = getBlobInfo(pLercBlob, blobSize, li, pMins, pMaxs, (size_t)nDepth * (size_t)nBands);
  if (e != kOk && isLerc1(pLercBlob, blobSize))
  {
    lerc_amd_context* h = threadHandle();
    if (!h) return kFailed;
    return lerc1BlobInfo(h->ctx, pLercBlob, blobSize, li, pMins, pMaxs, (size_t)nDepth * (size_t)nBands);
  }
  return e;
}

lerc_status lerc_decode_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks, unsigned char* pValidBytes,
  int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData, unsigned char* pUsesNoData,
  double* noDataValues)
{
  if (!pLercBlob || !blobSize || !pData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0)
    return kWrongParam;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return kWrongParam;
  return decodeHost(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, dataType, pData, false, pUsesNoData, noDataValues);
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
  if (!pLercBlob || !blobSize || !pData || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0) return kWrongParam;
  if (!masksArgOk(nMasks, nBands, pValidBytes)) return kWrongParam;
  BlobInfo li;
  u32 e = getBlobInfo(pLercBlob, blobSize, li);
  if (e != kOk && isLerc1(pLercBlob, blobSize))
  {
    lerc_amd_context* h = threadHandle();
    if (!h) return kFailed;
    e = lerc1BlobInfo(h->ctx, pLercBlob, blobSize, li, nullptr, nullptr, 0);
  }
  if (e != kOk) return e;
  if (li.nDepth != nDepth || li.nCols != nCols || li.nRows != nRows || li.nBands != nBands) return kFailed;
  return decodeHost(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, (unsigned)li.dt, pData, true, pUsesNoData,
                    noDataValues);
}

lerc_status lerc_decodeToDouble(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
  unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData)
{
  return lerc_decodeToDouble_4D(pLercBlob, blobSize, nMasks, pValidBytes, nDepth, nCols, nRows, nBands, pData, nullptr, nullptr);
}

// ---- device-pointer extension -------------------------------------------------------------------
lerc_amd_context* lerc_amd_create(void* hipStream)
{
  lerc_amd_context* h = new (std::nothrow) lerc_amd_context();
  if (!h) return nullptr;
  if (!h->ctx.ok()) { delete h; return nullptr; }
  h->ctx.setStream((hipStream_t)hipStream);
  return h;
}

void lerc_amd_destroy(lerc_amd_context* h) { delete h; }

void lerc_amd_set_stream(lerc_amd_context* h, void* hipStream) { if (h) h->ctx.setStream((hipStream_t)hipStream); }

const char* lerc_amd_last_error(lerc_amd_context* h)
{
  if (!h) h = threadHandle();    // nullptr: the calling thread's context behind the stock entry points
  return h ? h->ctx.lastError.c_str() : "no context";
}

lerc_status lerc_amd_encode_device(lerc_amd_context* h, const void* dData, unsigned int dataType, int nDepth, int nCols,
  int nRows, int nBands, int nMasks, const unsigned char* dValidBytes, double maxZErr, unsigned char* dOutBuffer,
  unsigned int outBufferSize, unsigned int* nBytesWritten)
{
  if (!h || !nBytesWritten) return kWrongParam;
  *nBytesWritten = 0;
  if (!dData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0) return kWrongParam;
  if (!masksArgOk(nMasks, nBands, dValidBytes)) return kWrongParam;
  if (!dimsOk(nDepth, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  EncodeRequest rq;
  rq.dData = dData; rq.dValidBytes = dValidBytes; rq.dt = (int)dataType; rq.nDepth = nDepth; rq.nCols = nCols; rq.nRows = nRows;
  rq.nBands = nBands; rq.nMasks = nMasks; rq.maxZErr = maxZErr; rq.dOut = dOutBuffer; rq.outCapacity = outBufferSize;
  u32 needed = 0, written = 0;
  const u32 rc = encodeDevice(h->ctx, rq, needed, written);
  if (rc == kOk) *nBytesWritten = dOutBuffer ? written : needed;
  return rc;
}

lerc_status lerc_amd_decode_device(lerc_amd_context* h, const unsigned char* dLercBlob, unsigned int blobSize, int nMasks,
  unsigned char* dValidBytes, int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* dData)
{
  if (!h || !dLercBlob || !blobSize || !dData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0)
    return kWrongParam;
  if (!masksArgOk(nMasks, nBands, dValidBytes)) return kWrongParam;
  if (!dimsOk(nDepth, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  DecodeRequest rq;
  rq.dBlob = dLercBlob; rq.blobSize = blobSize; rq.dt = (int)dataType; rq.nDepth = nDepth; rq.nCols = nCols; rq.nRows = nRows;
  rq.nBands = nBands; rq.nMasks = nMasks; rq.dOut = dData; rq.dValidBytes = dValidBytes;
  const u32 rc = decodeDevice(h->ctx, rq);
  if (rc == kFailed) wipeDecodeOutputs(h->ctx, rq);
  return rc;
}

// ---- asynchronous device API: operations queue up on the context's stream, the host runs ahead -------------------------
namespace {

// Brings every operation of the context to its result.  The stream is waited for once; verdicts are then read in enqueue
// order.  An operation the streaming kernels handed back (or that never was theirs) is repeated through the synchronous
// path -- and so is everything enqueued behind it, which may have worked on what it had not produced yet.
void completeAll(lerc_amd_context* h)
{
  bool pending = false;
  for (const AsyncOp& op : h->ops) pending = pending || !op.done;
  if (!pending) return;
  Context& ctx = h->ctx;
  const bool synced = ctx.sync();
  bool rerun = !synced;
  for (AsyncOp& op : h->ops)
  {
    if (op.done) continue;
    if (!rerun && op.streamed)
    {
      const u8* slot = ctx.asyncSlot(op.ticket);
      if (op.isEncode)
      {
        bool redo = false;
        u32 needed = 0, written = 0;
        encodeStreamingVerdict(ctx, op.er, slot, redo, op.status, needed, written);
        op.bytes = op.er.dOut ? written : needed;
        rerun = redo;
      }
      else rerun = !decodeStreamingVerdict(ctx, slot, op.epoch, nullptr, op.form, op.dr.dt);
      if (!rerun) { if (!op.isEncode) ctx.pathCount[2]++; op.done = true; continue; }
    }
    rerun = true;
    if (op.isEncode)
    {
      u32 needed = 0, written = 0;
      op.status = encodeDevice(ctx, op.er, needed, written);
      op.bytes = op.er.dOut ? written : needed;
    }
    else
    {
      // (the blob's true length: an asynchronous decode may have been given the capacity of the buffer an encode still
      // in flight was writing to; the synchronous path wants the blob's own size)
      if (op.dr.dBlob && !op.dr.hBlob && op.dr.blobSize >= 70)
      {
        u8 head[64];
        if (hipMemcpy(head, op.dr.dBlob, sizeof(head), hipMemcpyDeviceToHost) == hipSuccess)
        {
          BlobInfo info;
          if (getBlobInfo(head, sizeof(head), info) == kOk && info.blobSize >= 70 && info.blobSize <= op.dr.blobSize) op.dr.blobSize = info.blobSize;
        }
      }
      if (op.streamed) { op.dr.maxForm = op.form - 1; op.dr.noStreaming = op.dr.maxForm <= 0; }    // (that streaming form has just refused this blob)
      op.status = decodeDevice(ctx, op.dr);
      if (op.status == kFailed) wipeDecodeOutputs(ctx, op.dr);
    }
    op.done = true;
  }
}

unsigned pushOp(lerc_amd_context* h, AsyncOp& op)
{
  // results nobody asked for do not pile up: beyond the pinned slots' number the oldest are brought to their end and dropped
  if (h->ops.size() >= (size_t)Context::kAsyncSlots - 1) { completeAll(h); while (h->ops.size() >= (size_t)Context::kAsyncSlots / 2) h->ops.pop_front(); }
  op.ticket = h->nextTicket++;
  if (h->nextTicket == 0) h->nextTicket = 1;
  h->ops.push_back(op);
  return op.ticket;
}

}    // namespace

lerc_status lerc_amd_encode_device_async(lerc_amd_context* h, const void* dData, unsigned int dataType, int nDepth, int nCols,
  int nRows, int nBands, int nMasks, const unsigned char* dValidBytes, double maxZErr, unsigned char* dOutBuffer,
  unsigned int outBufferSize, unsigned int* ticket)
{
  if (!h || !ticket) return kWrongParam;
  *ticket = 0;
  if (!dData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0 || maxZErr < 0) return kWrongParam;
  if (!masksArgOk(nMasks, nBands, dValidBytes)) return kWrongParam;
  if (!dimsOk(nDepth, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  AsyncOp op;
  op.isEncode = true;
  EncodeRequest& rq = op.er;
  rq.dData = dData; rq.dValidBytes = dValidBytes; rq.dt = (int)dataType; rq.nDepth = nDepth; rq.nCols = nCols; rq.nRows = nRows;
  rq.nBands = nBands; rq.nMasks = nMasks; rq.maxZErr = maxZErr; rq.dOut = dOutBuffer; rq.outCapacity = outBufferSize;
  const unsigned t = pushOp(h, op);
  AsyncOp& q = h->ops.back();
  q.streamed = encodeEnqueueStreaming(h->ctx, q.er, h->ctx.asyncSlot(t));
  if (!q.streamed) completeAll(h);    // not a request the streaming kernels take: done on the spot (in order, behind what is in flight)
  *ticket = t;
  return kOk;
}

lerc_status lerc_amd_decode_device_async(lerc_amd_context* h, const unsigned char* dLercBlob, unsigned int blobSizeBound, int nMasks,
  unsigned char* dValidBytes, int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* dData, unsigned int* ticket)
{
  if (!h || !ticket) return kWrongParam;
  *ticket = 0;
  if (!dLercBlob || !blobSizeBound || !dData || dataType >= DT_Undefined || nDepth <= 0 || nCols <= 0 || nRows <= 0 || nBands <= 0)
    return kWrongParam;
  if (!masksArgOk(nMasks, nBands, dValidBytes)) return kWrongParam;
  if (!dimsOk(nDepth, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  AsyncOp op;
  DecodeRequest& rq = op.dr;
  rq.dBlob = dLercBlob; rq.blobSize = blobSizeBound; rq.dt = (int)dataType; rq.nDepth = nDepth; rq.nCols = nCols; rq.nRows = nRows;
  rq.nBands = nBands; rq.nMasks = nMasks; rq.dOut = dData; rq.dValidBytes = dValidBytes;
  const unsigned t = pushOp(h, op);
  AsyncOp& q = h->ops.back();
  q.streamed = decodeEnqueueStreaming(h->ctx, q.dr, h->ctx.asyncSlot(t), q.epoch);
  q.form = h->ctx.lastStreamForm;
  if (!q.streamed) completeAll(h);
  *ticket = t;
  return kOk;
}

lerc_status lerc_amd_finish(lerc_amd_context* h, unsigned int ticket, unsigned int* nBytes)
{
  if (!h) return kWrongParam;
  if (nBytes) *nBytes = 0;
  completeAll(h);
  if (ticket == 0) { h->ops.clear(); return kOk; }    // "everything": results are dropped
  for (auto it = h->ops.begin(); it != h->ops.end(); ++it)
    if (it->ticket == ticket)
    {
      const u32 rc = it->status;
      if (nBytes) *nBytes = it->bytes;
      h->ops.erase(it);
      return rc;
    }
  return kWrongParam;    // no such operation (handed out before, or dropped behind more than kAsyncSlots / 2 newer ones)
}

lerc_status lerc_amd_encode_tiles_device(lerc_amd_context* h, const void* dTiles, unsigned int dataType, int nCols, int nRows, int nTiles,
  double maxZErr, unsigned char* dArena, unsigned long long arenaCapacity, unsigned long long* offsets, unsigned int* sizes,
  unsigned long long* arenaUsed)
{
  if (!h || !dTiles || !dArena || !offsets || !sizes || dataType >= DT_Undefined || nCols <= 0 || nRows <= 0 || nTiles <= 0 || maxZErr < 0)
    return kWrongParam;
  if (!dimsOk(1, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  TilesEncodeRequest rq;
  rq.dData = dTiles; rq.dt = (int)dataType; rq.nCols = nCols; rq.nRows = nRows; rq.nTiles = nTiles; rq.maxZErr = maxZErr;
  rq.dArena = dArena; rq.arenaCapacity = arenaCapacity; rq.hOffsets = offsets; rq.hSizes = sizes;
  u64 used = 0;
  const u32 rc = encodeTilesDevice(h->ctx, rq, used);
  if (arenaUsed) *arenaUsed = used;
  return rc;
}

lerc_status lerc_amd_encode_tiles_device_slots(lerc_amd_context* h, const void* dTiles, unsigned int dataType, int nCols, int nRows, int nTiles,
  double maxZErr, unsigned char* dSlots, unsigned long long slotBytes, unsigned int* sizes)
{
  if (!h || !dTiles || !dSlots || !sizes || dataType >= DT_Undefined || nCols <= 0 || nRows <= 0 || nTiles <= 0 || maxZErr < 0
    || slotBytes == 0 || (slotBytes & 15u) != 0)
    return kWrongParam;
  if (!dimsOk(1, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  std::vector<u64> offsets((size_t)nTiles);
  TilesEncodeRequest rq;
  rq.dData = dTiles; rq.dt = (int)dataType; rq.nCols = nCols; rq.nRows = nRows; rq.nTiles = nTiles; rq.maxZErr = maxZErr;
  rq.dArena = dSlots; rq.arenaCapacity = (u64)nTiles * slotBytes; rq.hOffsets = offsets.data(); rq.hSizes = sizes; rq.slotBytes = slotBytes;
  u64 used = 0;
  return encodeTilesDevice(h->ctx, rq, used);
}

lerc_status lerc_amd_decode_tiles_device_slots(lerc_amd_context* h, const unsigned char* dSlots, unsigned long long slotBytes,
  const unsigned int* sizes, int nTiles, int nCols, int nRows, unsigned int dataType, void* dTiles)
{
  if (!h || !dSlots || !sizes || !dTiles || dataType >= DT_Undefined || nCols <= 0 || nRows <= 0 || nTiles <= 0 || slotBytes == 0 || (slotBytes & 15u) != 0)
    return kWrongParam;
  if (!dimsOk(1, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  std::vector<u64> offsets((size_t)nTiles);
  for (int t = 0; t < nTiles; t++) offsets[(size_t)t] = (u64)t * slotBytes;
  TilesDecodeRequest rq;
  rq.dArena = dSlots; rq.hOffsets = offsets.data(); rq.hSizes = sizes; rq.dt = (int)dataType; rq.nCols = nCols; rq.nRows = nRows; rq.nTiles = nTiles;
  rq.dOut = dTiles;
  return decodeTilesDevice(h->ctx, rq);
}

lerc_status lerc_amd_decode_tiles_device(lerc_amd_context* h, const unsigned char* dArena, const unsigned long long* offsets,
  const unsigned int* sizes, int nTiles, int nCols, int nRows, unsigned int dataType, void* dTiles)
{
  if (!h || !dArena || !offsets || !sizes || !dTiles || dataType >= DT_Undefined || nCols <= 0 || nRows <= 0 || nTiles <= 0) return kWrongParam;
  if (!dimsOk(1, nCols, nRows, (size_t)dtSize((int)dataType))) return kDimsTooLarge;
  TilesDecodeRequest rq;
  rq.dArena = dArena; rq.hOffsets = offsets; rq.hSizes = sizes; rq.dt = (int)dataType; rq.nCols = nCols; rq.nRows = nRows; rq.nTiles = nTiles;
  rq.dOut = dTiles;
  return decodeTilesDevice(h->ctx, rq);
}

void lerc_amd_profile_enable(lerc_amd_context* h, int on) { if (h) h->ctx.profEnable(on != 0); }

int lerc_amd_profile_read(lerc_amd_context* h, char* buf, int cap, int reset)
{
  if (!h || !buf || cap <= 0) return -1;
  const std::string r = h->ctx.profReport(reset != 0);
  const int n = (int)std::min<size_t>(r.size(), (size_t)cap - 1);
  memcpy(buf, r.data(), n);
  buf[n] = 0;
  return n;
}

const char* lerc_amd_last_note(lerc_amd_context* h)
{
  if (!h) h = threadHandle();
  return h ? h->ctx.lastNote.c_str() : "";
}

void lerc_amd_path_counters(lerc_amd_context* h, unsigned long long out[4])
{
  if (!h) h = threadHandle();    // the context behind the stock host-pointer entry points of this thread
  for (int i = 0; i < 4; i++) out[i] = h ? h->ctx.pathCount[i] : 0;
}

void lerc_amd_decode_forms(lerc_amd_context* h, unsigned long long out[4])
{
  if (!h) h = threadHandle();
  for (int i = 0; i < 4; i++) out[i] = h ? h->ctx.formCount[i] : 0;
}

unsigned int lerc_amd_gather_blobs(lerc_amd_context* h, void* ncclComm, int root, const void* dMessage, unsigned long long nBytes,
                                   void* dRootBuffer, unsigned long long rootCapacity, unsigned long long* hLengths, unsigned long long* hOffsets,
                                   void* stream)
{
  if (!h || !ncclComm) return kWrongParam;
  std::string err;
  const u32 rc = gatherBlobsRccl(ncclComm, root, dMessage, nBytes, dRootBuffer, rootCapacity, (u64*)hLengths, (u64*)hOffsets, (hipStream_t)stream, err);
  if (rc != kOk) h->ctx.lastError = err;
  return rc;
}

void lerc_amd_decode_refusals(lerc_amd_context* h, unsigned long long out[4])
{
  if (!h) h = threadHandle();
  for (int i = 0; i < 4; i++) out[i] = h ? h->ctx.refusalCount[i] : 0;
}

unsigned int lerc_amd_mask_rle_device(lerc_amd_context* h, const unsigned char* dBits, unsigned int nBytes, unsigned char* dOut,
                                      unsigned int cap, unsigned int* size)
{
  if (!h || !dBits || !dOut || !size || nBytes == 0 || ((uintptr_t)dBits & 15)) return kWrongParam;
  Context& ctx = h->ctx;
  ctx.reset();
  if (!ctx.reserve(maskRleScratchBytes(nBytes) + 4096)) return kFailed;
  u8* scratch = ctx.allocT<u8>(maskRleScratchBytes(nBytes));
  u32* dSize = ctx.allocT<u32>(4);
  u32* pin = (u32*)ctx.pinned(64);
  if (!scratch || !dSize || !pin) return kFailed;
  launchMaskRle(dBits, nBytes, dOut, cap, dSize, scratch, ctx.activeStream());
  hipMemcpyAsync(pin, dSize, 4, hipMemcpyDeviceToHost, ctx.activeStream());
  if (!ctx.sync()) return kFailed;
  if (pin[0] == 0xFFFFFFFFu) return kBufferTooSmall;
  *size = pin[0];
  return kOk;
}

unsigned int lerc_amd_mask_rle_decode_device(lerc_amd_context* h, const unsigned char* dRle, unsigned int rleBytes, unsigned char* dBits,
                                             unsigned int nBytes)
{
  if (!h || !dRle || !dBits || rleBytes < 2 || nBytes == 0) return kWrongParam;
  Context& ctx = h->ctx;
  ctx.reset();
  const size_t need = maskRleDecodeScratchBytes(rleBytes);
  if (!ctx.reserve(need + 4096)) return kFailed;
  u8* scratch = ctx.allocT<u8>(need);
  DeviceStatus* dStatus = reinterpret_cast<DeviceStatus*>(ctx.allocT<u8>(64));
  DeviceStatus* pin = (DeviceStatus*)ctx.pinned(64);
  if (!scratch || !dStatus || !pin) return kFailed;
  hipMemsetAsync(dStatus, 0, 64, ctx.activeStream());
  launchMaskRleDecode(dRle, rleBytes, dBits, nBytes, scratch, dStatus, ctx.activeStream());
  hipMemcpyAsync(pin, dStatus, sizeof(DeviceStatus), hipMemcpyDeviceToHost, ctx.activeStream());
  if (!ctx.sync()) return kFailed;
  return pin->error ? pin->error : kOk;
}

const char* lerc_amd_build_info(void)
{
#ifdef HIPSIM
  return "lerc_amd 0.1 hipsim (CPU SIMT emulator -- test build, not the product)";
#else
  return "lerc_amd 0.1 gfx950 hip";
#endif
}

}    // extern "C"
