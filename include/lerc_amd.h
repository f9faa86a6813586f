/*
 * lerc_amd.h -- C ABI of liblerc_amd.so, the MI355X-native LERC (Lerc2 v6) encode / decode path.
 *
 * Part 1 is the stock LERC C API: the twelve entry points below have exactly the names, argument
 * order, argument meaning and status codes of Esri/lerc's src/LercLib/include/Lerc_c_api.h, so a
 * host that binds libLerc (ctypes / P/Invoke / cgo / GDAL) binds this library unchanged
 * (INTEGRATION.md shows the bindings).  Every pointer is a HOST pointer owned by the caller; the
 * library stages through HBM, runs the HIP kernels and copies the result back.  There is no CPU
 * codec path inside: without a working HIP device every call returns lerc_status 1 (Failed) and
 * says so on stderr.
 *
 * Part 2, in lerc_amd_device.h, is new surface (not in the reference): the same codec on DEVICE
 * pointers, for callers that already hold rasters in HBM (bench.py, tile mosaics sharded over
 * several GPUs).
 *
 * Data layout (reference Lerc_c_api.h:113-124): raw pixels are row-major, top-left first,
 * [nBands][nRows][nCols][nDepth], little endian; masks are nMasks x nRows x nCols bytes, 1 = valid.
 * dataType: 0 char, 1 uchar, 2 short, 3 ushort, 4 int, 5 uint, 6 float, 7 double (Lerc_types.h:22-32).
 * lerc_status: 0 Ok, 1 Failed, 2 WrongParam, 3 BufferTooSmall, 4 NaN, 5 HasNoData,
 *              6 DimensionsTooLarge (Lerc_types.h:11-20).
 */
#ifndef LERC_AMD_H
#define LERC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define LERC_AMD_API __attribute__((visibility("default")))

/* Version of the stock API this library implements (reference Lerc_c_api.h:39-52: callers test features with
 * LERC_AT_LEAST_VERSION).  The shared object carries the reference's soname, libLerc.so.4 (CMakeLists.txt:27-29). */
#define LERC_VERSION_MAJOR 4
#define LERC_VERSION_MINOR 2
#define LERC_VERSION_PATCH 0
#define LERC_COMPUTE_VERSION(maj, min, patch) ((maj) * 10000 + (min) * 100 + (patch))
#define LERC_VERSION_NUMBER LERC_COMPUTE_VERSION(LERC_VERSION_MAJOR, LERC_VERSION_MINOR, LERC_VERSION_PATCH)
#define LERC_AT_LEAST_VERSION(maj, min, patch) (LERC_VERSION_NUMBER >= LERC_COMPUTE_VERSION(maj, min, patch))

typedef unsigned int lerc_status;

/* ------------------------------------------------------------------------------------------------
 * Part 1 -- stock API (replaces the reference implementation file src/LercLib/Lerc_c_api_impl.cpp)
 * ---------------------------------------------------------------------------------------------- */

/* reference Lerc_c_api.h:126-137 -- exact size lerc_encode() will write ("accurate to the byte") */
LERC_AMD_API lerc_status lerc_computeCompressedSize(const void* pData, unsigned int dataType, int nDepth, int nCols,
    int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes);

/* reference Lerc_c_api.h:141-154 -- zero-fills pOutBuffer, writes the blob, sets *nBytesWritten */
LERC_AMD_API lerc_status lerc_encode(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
    int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer,
    unsigned int outBufferSize, unsigned int* nBytesWritten);

/* reference Lerc_c_api.h:159-171 and :175-189 -- codecVersion -1 (latest) or 2..6; codec 2..5 blobs come out as
 * Lerc::EncodeInternal_v5 writes them (Lerc.cpp:526-624) */
LERC_AMD_API lerc_status lerc_computeCompressedSizeForVersion(const void* pData, int codecVersion, unsigned int dataType,
    int nDepth, int nCols, int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr,
    unsigned int* numBytes);
LERC_AMD_API lerc_status lerc_encodeForVersion(const void* pData, int codecVersion, unsigned int dataType, int nDepth,
    int nCols, int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr,
    unsigned char* pOutBuffer, unsigned int outBufferSize, unsigned int* nBytesWritten);

/* reference Lerc_c_api.h:203-210 -- header walk only, runs on the host (legacy Lerc1 "CntZImage" blobs: the bands are
 * decoded on the device to count valid pixels and find the range, as Lerc.cpp:184-266 does on the CPU).
 * infoArray: version, dataType, nDepth, nCols, nRows, nBands, nValidPixels(band 0), blobSize, nMasks,
 *            nDepth, nUsesNoDataValue;  dataRangeArray: zMin, zMax, maxZErrUsed (Lerc_types.h:34-56) */
LERC_AMD_API lerc_status lerc_getBlobInfo(const unsigned char* pLercBlob, unsigned int blobSize,
    unsigned int* infoArray, double* dataRangeArray, int infoArraySize, int dataRangeArraySize);

/* reference Lerc_c_api.h:221-228 */
LERC_AMD_API lerc_status lerc_getDataRanges(const unsigned char* pLercBlob, unsigned int blobSize, int nDepth,
    int nBands, double* pMins, double* pMaxs);

/* reference Lerc_c_api.h:238-252 -- pData / pValidBytes pre-allocated by the caller.  Lerc2 codec 2..6 and Lerc1 blobs
 * (Lerc1: pixels that are not valid keep what pData held, Lerc.cpp:2063-2107).
 * One difference to the reference when the call returns Failed: the reference checks a blob's checksum before it writes a pixel
 * (Lerc2.cpp:592-601) and leaves pData / pValidBytes as they were; here the streaming kernels write pixels while the checksum is
 * still being summed, so after a Failed decode of a blob they had taken up pData / pValidBytes hold ZEROS -- never pixels of a
 * blob that did not pass (the device-pointer calls of lerc_amd_device.h do the same to their device buffers).  Other statuses
 * (WrongParam, BufferTooSmall ...) are decided before anything is written.  (Behind a good checksum the reference, too, leaves a
 * partly written image when a block fails to parse.) */
LERC_AMD_API lerc_status lerc_decode(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData);

/* reference Lerc_c_api.h:258-270 */
LERC_AMD_API lerc_status lerc_decodeToDouble(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData);

/* reference Lerc_c_api.h:300-380 -- the _4D variants add a per-band noData value.  With
 * pUsesNoData == NULL (or all zero) they are the calls above.  A noData value turns pixels that hold it in every
 * depth into invalid ones and, for nDepth > 1, travels in the blob (remapped below the data range if need be);
 * lerc_decode_4D hands it back.  (Where the filter has to make a float band lossless, the band goes through the lossless float mode.) */
LERC_AMD_API lerc_status lerc_computeCompressedSize_4D(const void* pData, unsigned int dataType, int nDepth, int nCols,
    int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes,
    const unsigned char* pUsesNoData, const double* noDataValues);
LERC_AMD_API lerc_status lerc_encode_4D(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
    int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer,
    unsigned int outBufferSize, unsigned int* nBytesWritten, const unsigned char* pUsesNoData,
    const double* noDataValues);
LERC_AMD_API lerc_status lerc_decode_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData,
    unsigned char* pUsesNoData, double* noDataValues);
LERC_AMD_API lerc_status lerc_decodeToDouble_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData,
    unsigned char* pUsesNoData, double* noDataValues);

#ifdef __cplusplus
}
#endif

/* The device-pointer extension (no reference counterpart) lives in a header of its own; it is pulled in here so that
 * existing includes of lerc_amd.h keep compiling.  A host that only replaces libLerc needs nothing of it. */
#include "lerc_amd_device.h"

#endif
