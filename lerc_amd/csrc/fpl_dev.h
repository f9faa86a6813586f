// fpl_dev.h -- types and launch wrappers shared by fpl_host.cpp and fpl_kernels.hip (lossless float / double,
// Lerc2 image mode IEM_DeltaDeltaHuffman; reference: fpl_Lerc2Ext.cpp, fpl_UnitTypes.cpp, fpl_EsriHuffman.cpp).
#pragma once
#include "lerc_common.h"

namespace lerc {

static const int kFplPrime = 7;          // fpl_Compression.h:33 PRIME_MULT: the entropy estimates look at every 7th byte
static const int kFplMaxDelta = 5;       // fpl_Predictor.h:33 MAX_DELTA
static const int kFplLevels = kFplMaxDelta + 1;

// The band as the codec sees it (fpl_Lerc2Ext.cpp:432-452): nDepth == 1 -> cols x rows = nCols x nRows,
// otherwise cols = nDepth, rows = nCols * nRows.  unit = 4 (float) or 8 (double) bytes.
struct FplGeom
{
  i64 nElem, cols, rows;
  int unit;
  int nanToZero;     // nDepth == 1: a NaN in a valid pixel was set to 0 by the filter in front (Lerc.cpp:1439-1442)
};

inline i64 fplPlaneStride(i64 nElem) { return (nElem + 15) & ~(i64)15; }    // byte planes sit at 16-byte aligned distances
inline i64 fplColumnSegmentRows(i64 rows) { const i64 s = (rows + 4095) / 4096; return s < 64 ? 64 : s; }

struct FplSpan { i64 start, len; };      // a test block (in elements) resp. a snippet (in plane bytes)

struct FplLevels { int level[8]; };

// ---- encode
void launchFplPredictorSamples(const void* data, const u8* byteMask, const FplGeom& g, const FplSpan* blocks, u32 nBlocks,
                               u32* histos /* [nBlocks][3][2][unit][256] */, hipStream_t st);
void launchFplPredict(const void* data, const u8* byteMask, const FplGeom& g, int predictor, void* units, hipStream_t st);
void launchFplLevelSamples(const void* units, const FplGeom& g, const FplSpan* snippets, u32 nSnippets,
                           u32* histos /* [unit][nSnippets][kFplLevels][256] */, hipStream_t st);
void launchFplSymbols(const void* units, const FplGeom& g, const FplLevels& lv, u8* planes /* [unit][nElem] */,
                      u32* histos /* [unit][256] + [8]: bytes per plane that equal their successor; zeroed */, hipStream_t st);

// entropy estimates (fpl_Compression.cpp:85-113) of nHist byte histograms -> out[nHist].  log2Tables holds, for every
// histogram total t that occurs (total[k], from index at[k] on), log2((double)t / c) for c = 0 .. t as the host's libm
// gives them (index 0 unused); a histogram whose total has no table yields -1.
struct FplEntropyTables { u32 nTables; u32 total[4]; u32 at[4]; };
void launchFplEntropy(const u32* histos, u32 nHist, const double* log2Tables, const FplEntropyTables& tb, i64* out, hipStream_t st);

// PackBits (fpl_EsriHuffman.cpp:79-236) of one byte plane: size, then the stream itself
struct PackBitsBuffers { u32 *carry1, *carry2, *carry3; };    // packBitsCarryCount(n) words each
size_t packBitsCarryCount(u32 n);
void launchPackBitsPlan(const u8* plane, u32 n, const PackBitsBuffers& b, hipStream_t st);
const u32* packBitsSize(const PackBitsBuffers& b, u32 n);    // where the stream's size stands after launchPackBitsPlan
void launchPackBitsEmit(const u8* plane, u32 n, const PackBitsBuffers& b, u8* out, hipStream_t st);

// ---- decode
// PackBits stream -> plane; scratch holds packBitsDecodeScratchBytes(n) bytes (256-byte aligned), result = { bytes produced, ok }
size_t packBitsDecodeScratchBytes(u32 n);
bool launchPackBitsDecode(const u8* in, u32 n, u32 expected, u8* scratch, u32* result, u8* out, hipStream_t st);
// restoreSequence (fpl_Lerc2Ext.cpp:128-165), one level: p[i] += p[i - 1] for i = 1 .. n - 1
void launchBytePrefixSum(u8* p, u32 n, u32* scratch /* n / 1024 + 8 */, hipStream_t st);
// planes -> units -> predictor undone -> float bits back in place (fpl_Lerc2Ext.cpp:608-721)
void launchFplGather(const u8* planes, const int* byteIndex /* host, [unit] */, const FplGeom& g, bool finish, void* out, hipStream_t st);
void launchFplColumnSums(void* units, const FplGeom& g, void* partial /* [nSeg][cols] units */, u32 nSeg, hipStream_t st);
void launchFplRowSums(void* units, const FplGeom& g, hipStream_t st);    // also turns float units back into float bits
u32 fplColumnSegments(i64 rows);

}    // namespace lerc
