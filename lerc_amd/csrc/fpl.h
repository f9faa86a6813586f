// fpl.h -- lossless float / double image mode (IEM_DeltaDeltaHuffman; Lerc2.cpp:305-328, :674-678,
// fpl_Lerc2Ext.cpp): predictor + byte planes + per-plane entropy coding.  Kernels in fpl_kernels.hip, the
// decisions the reference takes from sampled entropy estimates on the host (fpl_host.cpp).
#pragma once
#include "codec.h"
#include "huffman.h"

namespace lerc {

struct FplPlanePlan
{
  int level = 0;        // extra byte-wise difference order of the plane (fpl_Lerc2Ext.cpp:569-571)
  int mode = 0;         // first byte of the plane's stream: 0 Huffman, 1 one value, 2 stored, 3 PackBits (fpl_EsriHuffman.cpp:240)
  u32 size = 0;         // bytes of the plane's stream, mode byte included
  u8 value = 0;         // mode 1
  HuffmanPlan huff;     // mode 0
};

struct FplPlan
{
  int predictor = 0;    // 0 none, 1 along rows, 2 rows and columns (fpl_Predictor.h:35)
  int unit = 4;
  i64 nElem = 0;
  int nRows = 0, nCols = 0, nDepth = 1;
  u8* dPlanes = nullptr;          // [unit] byte planes as they get entropy coded, fplPlaneStride() apart (context scratch)
  FplPlanePlan plane[8];
  u32 nBytes = 0;                 // LosslessFPCompression::compressedLength()
};

size_t fplEncodeScratchBytes(i64 nElem, int unit);
size_t fplDecodeScratchBytes(i64 nElem, int unit);

// statistics on the device, decisions on the host; false on a runtime error or when the reference would give up
// (nansFiltered: dData is the noData filter's private copy, NaNs of valid pixels are already replaced there; otherwise a NaN
// in a valid pixel of an nDepth == 1 band counts as 0, the way Lerc::FilterNoDataAndNaN leaves it, Lerc.cpp:1439-1442)
bool planLosslessFloat(Context& ctx, int dt, const void* dData, const u8* dByteMask, bool nansFiltered, int nRows, int nCols, int nDepth,
                       FplPlan& plan);
// writes plan.nBytes bytes at dOut
bool emitLosslessFloat(Context& ctx, const FplPlan& plan, u8* dOut);
// decodes the stream at band + dataBegin into dOut (all pixels, valid or not: fpl_Lerc2Ext.cpp:858); returns an ErrCode
u32 decodeLosslessFloat(Context& ctx, int dt, const u8* hBand, const u8* dBand, u32 dataBegin, u32 blobEnd, int nRows, int nCols,
                        int nDepth, void* dOut);

}    // namespace lerc
