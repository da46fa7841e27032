/*
 * scl_messages.h -- plain-C mirrors of the ROS interface of the loop-closure path and a ROS 1 wire codec.
 *
 * The reference talks to the other robots through three generated message types
 *   dlc_slam/global_descriptor          msg/global_descriptor.msg:2-8   (published at DM.h:1005-1024, consumed at DM.h:556-629)
 *   dlc_slam/loop_info                  msg/loop_info.msg:2-9           (filled at DM.h:1146-1158)
 *   dlc_slam/geometric_verification     srv/geometric_verification.srv:1-8 (request built at DM.h:1328-1333, response DM.h:1255-1260)
 * There is no ROS on the GPU box; these structs carry the same fields in the same order, and the codec produces /
 * parses the bytes roscpp's serializer would (little-endian; string and array fields prefixed with a uint32 length;
 * geometry_msgs/Transform = Vector3 + Quaternion of float64; Header = uint32 seq, time {uint32 sec, uint32 nsec},
 * string frame_id).  ros/msg and ros/srv hold the interface files.
 *
 * Ownership: decode functions do not copy arrays -- `values` / cloud `data` / strings point INTO the buffer that was
 * passed in and live as long as it does.  Every function returns SCL_OK or a negative scl_status
 * (SCL_ERR_INVALID_ARG for truncated / malformed input, SCL_ERR_NOMEM when the output buffer is too small; the
 * required size is still written to *len).
 */
#ifndef SCL_MESSAGES_H
#define SCL_MESSAGES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scl_msg_time      { uint32_t sec, nsec; } scl_msg_time;
typedef struct scl_msg_header    { uint32_t seq; scl_msg_time stamp; const char *frame_id; uint32_t frame_id_len; } scl_msg_header;
typedef struct scl_msg_vector3   { double x, y, z; } scl_msg_vector3;
typedef struct scl_msg_quaternion { double x, y, z, w; } scl_msg_quaternion;
typedef struct scl_msg_transform { scl_msg_vector3 translation; scl_msg_quaternion rotation; } scl_msg_transform;   /* geometry_msgs/Transform */

/* dlc_slam/global_descriptor (msg/global_descriptor.msg:2-8) */
typedef struct scl_msg_global_descriptor {
    scl_msg_header    header;
    int32_t           index;          /* keyframe index on the sending robot, DM.h:1021 */
    scl_msg_transform prePose;        /* DM.h:1011-1019 */
    scl_msg_transform curPose;        /* DM.h:1009 */
    const float      *values;         /* R*S floats, ring-major: what scl_make_and_save returns / scl_save_from_wire takes */
    uint32_t          n_values;
} scl_msg_global_descriptor;

/* dlc_slam/loop_info (msg/loop_info.msg:2-9) */
typedef struct scl_msg_loop_info {
    scl_msg_header    header;
    int32_t           robot0, robot1, index0, index1;   /* DM.h:1147-1150 */
    float             noise;                            /* ICP fitness score, DM.h:1151 */
    scl_msg_transform betPose;                          /* DM.h:1152-1158 */
} scl_msg_loop_info;

/* sensor_msgs/PointCloud2, as far as the path needs it: the record layout and where x, y, z sit */
typedef struct scl_msg_point_field { const char *name; uint32_t name_len; uint32_t offset; uint8_t datatype; uint32_t count; } scl_msg_point_field;
typedef struct scl_msg_cloud {
    scl_msg_header header;
    uint32_t height, width;
    const scl_msg_point_field *fields; uint32_t n_fields;     /* encode: caller's array; decode: up to 16 fields kept in the request struct */
    uint8_t  is_bigendian;
    uint32_t point_step, row_step;
    const uint8_t *data; uint32_t n_data;
    uint8_t  is_dense;
} scl_msg_cloud;

/* dlc_slam/geometric_verification request (srv/geometric_verification.srv:1-5) and response (:7-8) */
typedef struct scl_msg_geometric_verification_request {
    int32_t keyPre, keyCur, robotPre, robotCur;               /* DM.h:1328-1331 */
    scl_msg_cloud featureCloud;                               /* DM.h:1332-1333 */
    scl_msg_point_field field_store[16];                      /* decode target of featureCloud.fields */
} scl_msg_geometric_verification_request;
typedef struct scl_msg_geometric_verification_response {
    uint8_t           success;                                /* DM.h:1207, 1241, 1260 */
    scl_msg_transform poseBetween;                            /* DM.h:1255-1259 */
} scl_msg_geometric_verification_response;

/* ---- codec: *len receives the encoded size (also when buf is NULL or too small) ---------------------------------- */
int scl_msg_global_descriptor_encode(const scl_msg_global_descriptor *m, uint8_t *buf, size_t cap, size_t *len);
int scl_msg_global_descriptor_decode(const uint8_t *buf, size_t len, scl_msg_global_descriptor *m);
int scl_msg_loop_info_encode(const scl_msg_loop_info *m, uint8_t *buf, size_t cap, size_t *len);
int scl_msg_loop_info_decode(const uint8_t *buf, size_t len, scl_msg_loop_info *m);
int scl_msg_geometric_verification_request_encode(const scl_msg_geometric_verification_request *m, uint8_t *buf, size_t cap, size_t *len);
int scl_msg_geometric_verification_request_decode(const uint8_t *buf, size_t len, scl_msg_geometric_verification_request *m);
int scl_msg_geometric_verification_response_encode(const scl_msg_geometric_verification_response *m, uint8_t *buf, size_t cap, size_t *len);
int scl_msg_geometric_verification_response_decode(const uint8_t *buf, size_t len, scl_msg_geometric_verification_response *m);

/* The record layout pcl::toROSMsg gives a pcl::PointCloud<pcl::PointXYZI> (x, y, z float32 at 0 / 4 / 8, intensity at
 * 16, point_step 32): fills `cloud` so that it describes `n_points` records at `points` (borrowed). */
int scl_msg_cloud_from_xyzi(const void *points, uint32_t n_points, scl_msg_cloud *cloud, scl_msg_point_field fields_out[4]);
/* Where the x, y, z float32 fields of a decoded cloud sit: byte offsets inside a record.  The cloud comes from a peer, so the
 * layout is checked, not trusted: SCL_ERR_UNSUPPORTED when x, y, z are not three consecutive single float32 fields that end
 * inside the record (a duplicate x / y / z field, count != 1, another datatype, big-endian data, z past point_step), when
 * rows are padded (row_step != point_step * width), or when reading width * height whole records from data + xyz_offset --
 * what the engine's (pointer, count, stride) entry points do -- would pass the end of `data` (a record whose x is not at
 * offset 0 in a buffer without slack: repack it); SCL_ERR_INVALID_ARG when `data` is shorter than the records it declares. */
int scl_msg_cloud_xyz_layout(const scl_msg_cloud *cloud, int *stride_bytes, int *xyz_offset);

/* pose helpers of the path: geometry_msgs/Transform <-> (x, y, z, roll, pitch, yaw) with the conventions of
 * pcl::getTransformation / getTranslationAndEulerAngles / tf::createQuaternionMsgFromRollPitchYaw (DM.h:223, 1017, 1133) */
int scl_msg_transform_from_pose(double x, double y, double z, double roll, double pitch, double yaw, scl_msg_transform *t);
int scl_msg_transform_to_pose(const scl_msg_transform *t, double *x, double *y, double *z, double *roll, double *pitch, double *yaw);

#ifdef __cplusplus
}
#endif
#endif /* SCL_MESSAGES_H */
