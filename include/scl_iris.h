/*
 * scl_iris.h -- C ABI of the LiDAR-Iris descriptor (SURVEY.md 8(f)-4 part 2): the second descriptor of the reference
 * (class lidar_iris_descriptor, include/descriptor.h:462-1302; selected by config/dlc_lio_sam_params.yaml:23).
 *
 * On the GPU: the Iris image and row key of a scan (getIris, D.h:532-598), the log-Gabor binary templates T / M of an
 * image (logGaborFilter + logFeatureEncode, D.h:608-680), the Hamming matching of two keyframes' templates over column
 * shifts (getHammingDistance, D.h:932-964), the row-key distances of the candidate search, and on top of them the six
 * virtuals of the plugin (D.h:1026-1271: per-robot feature lists, local -> global index maps, intra- and inter-robot
 * detection).  The C++ adapter is include/scl/lidar_iris_hip_descriptor.hpp.
 *
 * The shift estimate in front of the matching (logPolarFFTTemplateMatch, D.h:793-925: forwardFFT, highpass, log-polar remap,
 * two cv::phaseCorrelate, warpAffine) is restated from the algorithms OpenCV publishes -- direct DFTs with fp64 sums, the
 * remaps in OpenCV's fixed point, the 5 x 5 weighted centroid -- and compare() (D.h:964-1024) evaluates the Hamming distance
 * in the windows of five column shifts around it, for the candidate as it is and turned by 180 columns, as match_num says:
 * the detections follow the reference's own procedure (scl_iris_fft_match, scl_iris_compare expose the steps).  What the
 * restatement cannot reproduce is OpenCV's float rounding (its mixed-radix FFT, the packed spectra of phaseCorrelate's
 * helpers): an estimate that OpenCV places within rounding of a whole number can fall on the other side here.
 * cfg.shift_search = 1 searches every column shift instead (round 2's behaviour; an opt-in, see the field).
 * Parity: bit-identical to the CPU restatement under oracle/ (tests/test_iris.py, tests/test_iris_fftmatch.py); against the
 * reference's binaries templates and estimates are UNPINNED (no OpenCV in the image) -- see oracle/iris_oracle.h.
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
    /* the plugin layer */
    double dist_thres;          /* 0.32: loop accepted below it (D.h:1140, 1245)            */
    int    num_exclude_recent;  /* 30:   newest keyframes of this robot kept out (D.h:1097)  */
    int    match_num;           /* 2:    which passes compare() runs (D.h:968-1021): 2 both, 1 only the candidate turned by
                                         180 columns, 0 only the first                                                */
    int    num_candidates;      /* 10:   row-key neighbours compared (D.h:1109)              */
    int    robot_num;           /* 1  */
    int    this_id;             /* 0  */
    float  knn_exclude_eps;     /* FLT_EPSILON: libnabo's knn with optionFlags = 0 skips neighbours whose squared row-key
                                   distance is <= this (no self match), D.h:1109, 1215; 0 = every key counts          */
    int    wire_decode;         /* 0: saveDescriptorAndKey's own indexing iris[row*(cols+1)+col+1] (D.h:1030-1037: reads
                                      the image sheared by one more column per row, stays inside the buffer);
                                   1: the layout makeAndSaveDescriptorAndKey emits (row*cols+col, D.h:1067-1074)       */
    int    shift_search;        /* 0 (default): compare() as the reference runs it (D.h:964-1024) -- the FFT shift estimate
                                      logPolarFFTTemplateMatch (D.h:793-925), then the Hamming windows of five column shifts
                                      around it, both passes as match_num says (even rows / cols);
                                   1: EVERY column shift instead of the estimate + windows (an opt-in: the distance can only
                                      be smaller than the reference's, so dist_thres 0.32 accepts loops the reference would
                                      not; the returned shift is the first minimum over [0, cols)) */
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
/* save (D.h:1046-1060) from an image and its row key, e.g. decoded from the wire by the caller */
int  scl_iris_save_image(scl_iris *h, const uint8_t *image, const float *rowkey, int8_t robot, int index);
/* saveDescriptorAndKey(const float*), D.h:1026-1044: `values` = rows*cols + rows floats as emitted by
 * scl_iris_make_and_save / the reference's makeAndSaveDescriptorAndKey; decoded as cfg.wire_decode says (float ->
 * uint8 like the reference's implicit conversion on x86: truncation, then the low 8 bits). */
int  scl_iris_save_from_wire(scl_iris *h, const float *values, int8_t robot, int index);
/* getSize(idIn), D.h:1260-1270: id = -1 -> keyframes of all robots, else those of robot `id` */
int  scl_iris_get_size(const scl_iris *h);
int  scl_iris_get_size_of(const scl_iris *h, int id);
/* getIndex(key), D.h:1255-1258: global key -> (robot, index) */
int  scl_iris_get_index(const scl_iris *h, int key, int8_t *robot, int *index);
/* global key of robot `robot`'s local keyframe `local` (local2Global, D.h:1055) */
int  scl_iris_local_to_global(const scl_iris *h, int robot, int local, int *key);
/* detectIntraLoopClosureID(curPtr), D.h:1085-1151: cur = LOCAL index among this_id's keyframes; candidates = the
 * num_candidates nearest row keys among this robot's keyframes [0, cur - num_exclude_recent); *loop_id = LOCAL index
 * of the best candidate if its distance < dist_thres, else -1; *bias = its column shift; *dist = the smallest
 * distance seen (10000000 if none), loop or not. */
int  scl_iris_detect_intra(scl_iris *h, int cur, int *loop_id, float *bias, float *dist);
/* detectInterLoopClosureID(curPtr), D.h:1153-1253: cur = GLOBAL key; a keyframe of this robot is searched among all
 * other robots' keyframes, a received one among this robot's; *loop_id = GLOBAL key or -1. */
int  scl_iris_detect_inter(scl_iris *h, int cur, int *loop_id, float *bias, float *dist);
/* the stored image / row key / templates of keyframe `key`: T and M are (2*nscale*rows) x cols bytes, 0 or 255, in the
 * row order of cv::vconcat at D.h:669-678 (real parts of the scales, then imaginary parts) */
int  scl_iris_get_image(scl_iris *h, int key, uint8_t *image, float *rowkey);
int  scl_iris_get_feature(scl_iris *h, int key, uint8_t *T, uint8_t *M);
/* getHammingDistance(T1, M1, T2, M2, scale), D.h:932-964, for keyframes key1 (shifted) and key2: the five column shifts
 * scale-2 .. scale+2; dis = NaN and bias = -1 when no shift has an unmasked bit. */
int  scl_iris_hamming(scl_iris *h, int key1, int key2, int scale, float *dis, int *bias);
/* key1 against n candidates, each with its own shift estimate */
int  scl_iris_hamming_batch(scl_iris *h, int key1, const int *cand, const int *scales, int n, float *dis, int *bias);
/* fftMatch(im0, im1) (D.h:927-932) with im0 = the image of key0 turned by roll0 columns (0, or 180 for compare()'s second pass)
 * and im1 = the image of key1: *center_x = the RotatedRect's centre x as a float (compare() takes int(center_x - cols / 2) as its
 * shift); *compatible (may be NULL) = 0 where the reference reports "Images are not compatible" (centre 0). */
int  scl_iris_fft_match(scl_iris *h, int key0, int roll0, int key1, float *center_x, int *compatible);
/* compare(key1, cand[i], &bias) for n candidates (D.h:964-1024, match_num as configured): distance (NaN where every window shift
 * is fully masked) and shift */
int  scl_iris_compare(scl_iris *h, int key1, const int *cand, int n, float *dis, int *bias);
/* every column shift 0 .. cols-1 (first minimum): what cfg.shift_search = 1 uses */
int  scl_iris_hamming_all_shifts(scl_iris *h, int key1, const int *cand, int n, float *dis, int *bias);

#ifdef __cplusplus
}
#endif
#endif /* SCL_IRIS_H */
