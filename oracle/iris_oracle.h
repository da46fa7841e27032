/* iris_oracle.h -- CPU restatement of the LiDAR-Iris building blocks (TEST INFRASTRUCTURE ONLY; nothing under
 * scl_slam_amd/ or include/ links, imports or executes this).
 *
 * Follows reference include/descriptor.h: getIris (D.h:532-598), logGaborFilter / logFeatureEncode (D.h:608-680),
 * circShift (D.h:600-606 and 566-598), getHammingDistance (D.h:932-964), the wire layout of makeAndSave /
 * saveDescriptorAndKey (D.h:1026-1083).  OpenCV is absent: cv::dft / cv::idft are restated as the direct fp64 DFT
 * they approximate (forward unscaled, inverse unscaled -- cv::idft without DFT_SCALE), cv::log / pow / exp on Mat1f as
 * float arithmetic.  PARITY UNPINNED against the reference's own binaries (OpenCV's float FFT rounds differently: a
 * template bit can differ where the filter response is within ~1e-3 relative of zero); the shift estimate
 * logPolarFFTTemplateMatch (D.h:793-925: cv::dft, remap, phaseCorrelate, warpAffine) is restated from the algorithms OpenCV
 * publishes (iriso_fft_match), equally unpinned.
 */
#ifndef IRIS_ORACLE_H
#define IRIS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct iriso_config {
    int rows, cols, nscan, nscale, min_wavelength;     /* 80, 360, 64, 4, 18  (D.h:474-483) */
    float mult, sigma_onf;                             /* 1.6, 0.75 */
} iriso_config;

/* getIris, D.h:532-598: image rows*cols bytes (row-major), rowkey rows floats */
/* glibc's float atan2f restated (fp32 ops; equal to libm on 4e9 pairs: oracle/tools/atan2f_check.c) -- std::atan2(float, float) of D.h:547-549 */
float iriso_atan2f(float y, float x);
void iriso_make_image(const iriso_config *c, const void *pts, int n, int stride_bytes, uint8_t *image, float *rowkey);
/* logFeatureEncode, D.h:661-680: T and M, (2*nscale*rows) x cols bytes each, 0 / 255 */
void iriso_encode(const iriso_config *c, const uint8_t *image, uint8_t *T, uint8_t *M);
/* the filter responses themselves (tests: how close to zero are the bits that decide), scale-major: [nscale][rows][cols][2] doubles */
void iriso_responses(const iriso_config *c, const uint8_t *image, double *resp);
/* getHammingDistance, D.h:932-964: shifts scale-2 .. scale+2 */
void iriso_hamming(const iriso_config *c, const uint8_t *T1, const uint8_t *M1, const uint8_t *T2, const uint8_t *M2, int scale, float *dis, int *bias);
/* every column shift 0 .. cols-1 (what the FFT estimate of D.h:793-925 narrows down): first minimum */
void iriso_hamming_all(const iriso_config *c, const uint8_t *T1, const uint8_t *M1, const uint8_t *T2, const uint8_t *M2, float *dis, int *bias);

/* fftMatch(im0, im1) = logPolarFFTTemplateMatch on copies (D.h:793-932), restated from OpenCV's published algorithms (see the
 * .c file: PARITY UNPINNED): *center_x = the RotatedRect's centre x as a float; returns 1, 0 for "images are not compatible"
 * (centre 0), -1 for sizes this restatement does not take (odd).  dbg (optional, 6 doubles): rotation_and_scale.x / .y, angle,
 * scale, tr.x, tr.y */
int iriso_fft_match(int rows, int cols, const uint8_t *im0, const uint8_t *im1, float *center_x, double *dbg);
/* compare(img1, img2, &bias), D.h:964-1024: the Hamming windows of five shifts around the FFT estimate(s) */
void iriso_compare(const iriso_config *c, int match_num, const uint8_t *img1, const uint8_t *T1, const uint8_t *M1,
                   const uint8_t *img2, const uint8_t *T2, const uint8_t *M2, float *dis, int *bias, int *shifts_out);

#ifdef __cplusplus
}
#endif
#endif
