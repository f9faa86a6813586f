/*
 * oracle/lerc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (sequential, single-threaded C++) of the Esri/lerc Lerc2 v6 codec path that the
 * MI355X product accelerates.  It exists only so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check the HIP path; nothing under lerc_amd/ may include, link or
 * dlopen it.
 *
 * The exported C symbols deliberately carry the same names and signatures as the reference C API
 * (reference: src/LercLib/include/Lerc_c_api.h:126-380) so one ctypes harness can drive the real
 * reference build (oracle/_ref/libLercRef.so), this restatement and the product library
 * interchangeably.  Always load it with RTLD_LOCAL (ctypes.CDLL default).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py byte-compares this restatement with
 * oracle/_ref (the real reference compiled from /root/reference) over a dtype x shape x maxZError x
 * data-shape matrix, and tests/test_golden.py checks it against the committed golden vectors
 * (doc/MORE.md worked example, JS sanity blob, testData digests).
 *
 * Lossless float / double is restated too (fpl_*); there the reference leaves the read-ahead word behind every
 * Huffman coded byte plane uninitialised (heap garbage), this restatement writes 0 (tests/cases.py: lossless_float_dont_care).
 *
 * Legacy Lerc1 ("CntZImage") blobs are restated too (decode only, like the reference) and pinned on testData/world.lerc1.
 */
#ifndef LERC_ORACLE_H
#define LERC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned int lerc_status;

#define ORC_API __attribute__((visibility("default")))

ORC_API lerc_status lerc_computeCompressedSize(const void* pData, unsigned int dataType, int nDepth, int nCols,
    int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes);

ORC_API lerc_status lerc_encode(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
    int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer,
    unsigned int outBufferSize, unsigned int* nBytesWritten);

ORC_API lerc_status lerc_computeCompressedSizeForVersion(const void* pData, int codecVersion, unsigned int dataType,
    int nDepth, int nCols, int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr,
    unsigned int* numBytes);

ORC_API lerc_status lerc_encodeForVersion(const void* pData, int codecVersion, unsigned int dataType, int nDepth,
    int nCols, int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr,
    unsigned char* pOutBuffer, unsigned int outBufferSize, unsigned int* nBytesWritten);

ORC_API lerc_status lerc_getBlobInfo(const unsigned char* pLercBlob, unsigned int blobSize, unsigned int* infoArray,
    double* dataRangeArray, int infoArraySize, int dataRangeArraySize);

ORC_API lerc_status lerc_getDataRanges(const unsigned char* pLercBlob, unsigned int blobSize, int nDepth, int nBands,
    double* pMins, double* pMaxs);

ORC_API lerc_status lerc_decode(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData);

ORC_API lerc_status lerc_decodeToDouble(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData);

ORC_API lerc_status lerc_computeCompressedSize_4D(const void* pData, unsigned int dataType, int nDepth, int nCols,
    int nRows, int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned int* numBytes,
    const unsigned char* pUsesNoData, const double* noDataValues);

ORC_API lerc_status lerc_encode_4D(const void* pData, unsigned int dataType, int nDepth, int nCols, int nRows,
    int nBands, int nMasks, const unsigned char* pValidBytes, double maxZErr, unsigned char* pOutBuffer,
    unsigned int outBufferSize, unsigned int* nBytesWritten, const unsigned char* pUsesNoData,
    const double* noDataValues);

ORC_API lerc_status lerc_decode_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, unsigned int dataType, void* pData,
    unsigned char* pUsesNoData, double* noDataValues);

ORC_API lerc_status lerc_decodeToDouble_4D(const unsigned char* pLercBlob, unsigned int blobSize, int nMasks,
    unsigned char* pValidBytes, int nDepth, int nCols, int nRows, int nBands, double* pData,
    unsigned char* pUsesNoData, double* noDataValues);

/* Extra probes used by the tests (not part of the reference API). */
ORC_API unsigned int orc_fletcher32(const unsigned char* bytes, int len);
/* Walks the tiling payload of a single-band blob and reports, per micro-block in stream order,
 * its byte offset (relative to blob start) and block flag byte.  Returns the number of blocks, or
 * a negative value if the blob is not in tiling mode / cannot be parsed. */
ORC_API long long orc_blockTable(const unsigned char* blob, unsigned int blobSize, unsigned int* offsets,
    unsigned char* flags, long long capacity);

#ifdef __cplusplus
}
#endif
#endif
