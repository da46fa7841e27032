/*
 * scl_iris.h -- C ABI of the LiDAR-Iris building blocks (SURVEY.md 8(f)-4 part 2): the second descriptor of the
 * reference (class lidar_iris_descriptor, include/descriptor.h:462-1302; selected by config/dlc_lio_sam_params.yaml:23).
 *
 * On the GPU: the Iris image and row key of a scan (getIris, D.h:532-598), the log-Gabor binary templates T / M of an
 * image (logGaborFilter + logFeatureEncode, D.h:608-680), and the Hamming matching of two keyframes' templates over
 * column shifts (getHammingDistance, D.h:932-964).  NOT here: the shift estimate in front of the matching
 * (logPolarFFTTemplateMatch, D.h:793-925 -- a chain of OpenCV calls whose arithmetic cannot be restated bit for bit
 * without OpenCV); scl_iris_hamming takes the estimate as an argument, scl_iris_hamming_all_shifts searches every
 * column shift instead (a superset of the reference's +-2 window around the estimate).
 * Parity: bit-identical to the CPU restatement under oracle/ (tests/test_gpu_iris.py); against the reference's binaries
 * the templates are unpinned (OpenCV's float FFT) -- see oracle/iris_oracle.h.
 * Conventions as in scl_engine.h (status codes, point clouds as pointer / count / stride, no CPU fallback).
 */
#ifndef SCL_IRIS_H
#define SCL_IRIS_H

#include <stdint.h>

#include "scl_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scl_iris scl_iris;

/* constructor arguments of lidar_iris_descriptor, D.h:473-486 (same defaults) */
typedef struct scl_iris_config {
    int    rows;                /* 80   */
    int    cols;                /* 360  */
    int    nscan;               /* 64 (or 16) */
    int    nscale;              /* 4    */
    int    min_wavelength;      /* 18   */
    float  mult;                /* 1.6  */
    float  sigma_onf;           /* 0.75 */
    int    device;
} scl_iris_config;

int  scl_iris_default_config(scl_iris_config *cfg);
int  scl_iris_create(const scl_iris_config *cfg, scl_iris **out);
int  scl_iris_destroy(scl_iris *h);
const char *scl_iris_last_error(const scl_iris *h);

/* getIris, D.h:532-598: image = rows*cols bytes (row-major: distance bin, yaw bin; bit q = elevation bin q seen),
 * rowkey = rows floats (row means of the per-cell maximum height).  Nothing is stored. */
int  scl_iris_make_image(scl_iris *h, const void *points, int n_points, int stride_bytes, uint8_t *image, float *rowkey);
/* makeAndSaveDescriptorAndKey, D.h:1062-1083: image + row key + templates are built and appended to the database;
 * out_values (rows*cols + rows floats, may be NULL) = the vector the reference returns (image values row-major, then
 * the row key). */
int  scl_iris_make_and_save(scl_iris *h, const void *points, int n_points, int stride_bytes, int8_t robot, int index, float *out_values);
/* save (D.h:1046-1060) from an image and its row key, e.g. decoded from the wire by the caller.  (The reference's own
 * decoder, D.h:1026-1044, reads iris[row*(cols+1)+col+1] from a buffer laid out with stride cols -- a defect that
 * shears the image; it is not replicated.) */
int  scl_iris_save_image(scl_iris *h, const uint8_t *image, const float *rowkey, int8_t robot, int index);
int  scl_iris_get_size(const scl_iris *h);
int  scl_iris_get_index(const scl_iris *h, int key, int8_t *robot, int *index);
/* the stored image / row key / templates of keyframe `key`: T and M are (2*nscale*rows) x cols bytes, 0 or 255, in the
 * row order of cv::vconcat at D.h:669-678 (real parts of the scales, then imaginary parts) */
int  scl_iris_get_image(scl_iris *h, int key, uint8_t *image, float *rowkey);
int  scl_iris_get_feature(scl_iris *h, int key, uint8_t *T, uint8_t *M);
/* getHammingDistance(T1, M1, T2, M2, scale), D.h:932-964, for keyframes key1 (shifted) and key2: the five column shifts
 * scale-2 .. scale+2; dis = NaN and bias = -1 when no shift has an unmasked bit. */
int  scl_iris_hamming(scl_iris *h, int key1, int key2, int scale, float *dis, int *bias);
/* key1 against n candidates, each with its own shift estimate */
int  scl_iris_hamming_batch(scl_iris *h, int key1, const int *cand, const int *scales, int n, float *dis, int *bias);
/* every column shift 0 .. cols-1 (first minimum): stands in for estimate + window where no estimate is available */
int  scl_iris_hamming_all_shifts(scl_iris *h, int key1, const int *cand, int n, float *dis, int *bias);

#ifdef __cplusplus
}
#endif
#endif /* SCL_IRIS_H */
