// messages.hip -- ROS 1 wire codec of the loop-closure path's messages (include/scl_messages.h).  Host code only.
//
// roscpp serialisation rules used here: little-endian scalars; `string` and variable-length arrays carry a uint32
// element count first; `time` = uint32 sec + uint32 nsec; `bool` = one byte; nested messages are laid out field by
// field in declaration order (Header: seq, stamp, frame_id; geometry_msgs/Transform: Vector3 x y z, Quaternion x y z w,
// all float64; sensor_msgs/PointCloud2: header, height, width, PointField[] fields {string name, uint32 offset,
// uint8 datatype, uint32 count}, bool is_bigendian, uint32 point_step, uint32 row_step, uint8[] data, bool is_dense).
#include "scl_messages.h"

#include <cmath>
#include <cstring>

#include "scl_engine.h"

namespace {

struct Writer {
    uint8_t *buf; size_t cap; size_t pos = 0;
    Writer(uint8_t *b, size_t c) : buf(b), cap(c) {}
    void raw(const void *p, size_t n) { if (buf && pos + n <= cap && n) std::memcpy(buf + pos, p, n); pos += n; }
    void u8(uint8_t v) { raw(&v, 1); }
    void u32(uint32_t v) { raw(&v, 4); }
    void i32(int32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); }
    void f64(double v) { raw(&v, 8); }
    void str(const char *s, uint32_t n) { u32(n); raw(s, n); }
    void header(const scl_msg_header &h) { u32(h.seq); u32(h.stamp.sec); u32(h.stamp.nsec); str(h.frame_id, h.frame_id ? h.frame_id_len : 0); }
    void transform(const scl_msg_transform &t)
    {
        f64(t.translation.x); f64(t.translation.y); f64(t.translation.z);
        f64(t.rotation.x); f64(t.rotation.y); f64(t.rotation.z); f64(t.rotation.w);
    }
    int finish(size_t *len) const
    {
        if (len) *len = pos;
        if (!buf) return SCL_OK;                             // size query
        return pos <= cap ? SCL_OK : SCL_ERR_NOMEM;
    }
};

struct Reader {
    const uint8_t *buf; size_t len; size_t pos = 0; bool ok = true;
    Reader(const uint8_t *b, size_t n) : buf(b), len(n) {}
    const uint8_t *take(size_t n)
    {
        if (!ok || n > len - pos) { ok = false; return nullptr; }
        const uint8_t *p = buf + pos; pos += n; return p;
    }
    template <class T> T scalar() { T v{}; const uint8_t *p = take(sizeof(T)); if (p) std::memcpy(&v, p, sizeof(T)); return v; }
    void str(const char **s, uint32_t *n) { *n = scalar<uint32_t>(); *s = reinterpret_cast<const char *>(take(*n)); if (!ok) { *s = nullptr; *n = 0; } }
    void header(scl_msg_header *h) { h->seq = scalar<uint32_t>(); h->stamp.sec = scalar<uint32_t>(); h->stamp.nsec = scalar<uint32_t>(); str(&h->frame_id, &h->frame_id_len); }
    void transform(scl_msg_transform *t)
    {
        t->translation.x = scalar<double>(); t->translation.y = scalar<double>(); t->translation.z = scalar<double>();
        t->rotation.x = scalar<double>(); t->rotation.y = scalar<double>(); t->rotation.z = scalar<double>(); t->rotation.w = scalar<double>();
    }
    int finish() const { return ok && pos == len ? SCL_OK : SCL_ERR_INVALID_ARG; }   // trailing bytes are an error too
};

void write_cloud(Writer &w, const scl_msg_cloud &c)
{
    w.header(c.header);
    w.u32(c.height); w.u32(c.width);
    w.u32(c.fields ? c.n_fields : 0);
    for (uint32_t i = 0; c.fields && i < c.n_fields; ++i) {
        const scl_msg_point_field &f = c.fields[i];
        w.str(f.name, f.name ? f.name_len : 0); w.u32(f.offset); w.u8(f.datatype); w.u32(f.count);
    }
    w.u8(c.is_bigendian); w.u32(c.point_step); w.u32(c.row_step);
    w.u32(c.data ? c.n_data : 0); w.raw(c.data, c.data ? c.n_data : 0);
    w.u8(c.is_dense);
}

void read_cloud(Reader &r, scl_msg_cloud *c, scl_msg_point_field *store, uint32_t store_cap)
{
    r.header(&c->header);
    c->height = r.scalar<uint32_t>(); c->width = r.scalar<uint32_t>();
    const uint32_t nf = r.scalar<uint32_t>();
    c->fields = store; c->n_fields = 0;
    for (uint32_t i = 0; i < nf && r.ok; ++i) {
        scl_msg_point_field f{};
        r.str(&f.name, &f.name_len); f.offset = r.scalar<uint32_t>(); f.datatype = r.scalar<uint8_t>(); f.count = r.scalar<uint32_t>();
        if (c->n_fields < store_cap) store[c->n_fields++] = f;
    }
    c->is_bigendian = r.scalar<uint8_t>(); c->point_step = r.scalar<uint32_t>(); c->row_step = r.scalar<uint32_t>();
    c->n_data = r.scalar<uint32_t>(); c->data = r.take(c->n_data);
    if (!r.ok) { c->data = nullptr; c->n_data = 0; }
    c->is_dense = r.scalar<uint8_t>();
}

}  // namespace

extern "C" {

int scl_msg_global_descriptor_encode(const scl_msg_global_descriptor *m, uint8_t *buf, size_t cap, size_t *len)
{
    if (!m || (m->n_values && !m->values)) return SCL_ERR_INVALID_ARG;
    Writer w(buf, cap);
    w.header(m->header); w.i32(m->index); w.transform(m->prePose); w.transform(m->curPose);
    w.u32(m->n_values); w.raw(m->values, sizeof(float) * (size_t)m->n_values);
    return w.finish(len);
}

int scl_msg_global_descriptor_decode(const uint8_t *buf, size_t len, scl_msg_global_descriptor *m)
{
    if (!buf || !m) return SCL_ERR_INVALID_ARG;
    Reader r(buf, len);
    r.header(&m->header); m->index = r.scalar<int32_t>(); r.transform(&m->prePose); r.transform(&m->curPose);
    m->n_values = r.scalar<uint32_t>();
    const uint8_t *p = (uint64_t)m->n_values * 4 <= len ? r.take(sizeof(float) * (size_t)m->n_values) : (r.ok = false, nullptr);
    m->values = reinterpret_cast<const float *>(p);           // (ROS buffers are not float-aligned in general: callers that need
    if (!r.ok) { m->values = nullptr; m->n_values = 0; }      //  alignment copy; the engine's scl_save_from_wire memcpy's anyway)
    return r.finish();
}

int scl_msg_loop_info_encode(const scl_msg_loop_info *m, uint8_t *buf, size_t cap, size_t *len)
{
    if (!m) return SCL_ERR_INVALID_ARG;
    Writer w(buf, cap);
    w.header(m->header); w.i32(m->robot0); w.i32(m->robot1); w.i32(m->index0); w.i32(m->index1); w.f32(m->noise); w.transform(m->betPose);
    return w.finish(len);
}

int scl_msg_loop_info_decode(const uint8_t *buf, size_t len, scl_msg_loop_info *m)
{
    if (!buf || !m) return SCL_ERR_INVALID_ARG;
    Reader r(buf, len);
    r.header(&m->header);
    m->robot0 = r.scalar<int32_t>(); m->robot1 = r.scalar<int32_t>(); m->index0 = r.scalar<int32_t>(); m->index1 = r.scalar<int32_t>();
    m->noise = r.scalar<float>(); r.transform(&m->betPose);
    return r.finish();
}

int scl_msg_geometric_verification_request_encode(const scl_msg_geometric_verification_request *m, uint8_t *buf, size_t cap, size_t *len)
{
    if (!m) return SCL_ERR_INVALID_ARG;
    Writer w(buf, cap);
    w.i32(m->keyPre); w.i32(m->keyCur); w.i32(m->robotPre); w.i32(m->robotCur);
    write_cloud(w, m->featureCloud);
    return w.finish(len);
}

int scl_msg_geometric_verification_request_decode(const uint8_t *buf, size_t len, scl_msg_geometric_verification_request *m)
{
    if (!buf || !m) return SCL_ERR_INVALID_ARG;
    Reader r(buf, len);
    m->keyPre = r.scalar<int32_t>(); m->keyCur = r.scalar<int32_t>(); m->robotPre = r.scalar<int32_t>(); m->robotCur = r.scalar<int32_t>();
    read_cloud(r, &m->featureCloud, m->field_store, 16);
    return r.finish();
}

int scl_msg_geometric_verification_response_encode(const scl_msg_geometric_verification_response *m, uint8_t *buf, size_t cap, size_t *len)
{
    if (!m) return SCL_ERR_INVALID_ARG;
    Writer w(buf, cap);
    w.u8(m->success ? 1 : 0); w.transform(m->poseBetween);
    return w.finish(len);
}

int scl_msg_geometric_verification_response_decode(const uint8_t *buf, size_t len, scl_msg_geometric_verification_response *m)
{
    if (!buf || !m) return SCL_ERR_INVALID_ARG;
    Reader r(buf, len);
    m->success = r.scalar<uint8_t>(); r.transform(&m->poseBetween);
    return r.finish();
}

int scl_msg_cloud_from_xyzi(const void *points, uint32_t n_points, scl_msg_cloud *cloud, scl_msg_point_field fields_out[4])
{
    if (!cloud || !fields_out || (n_points && !points)) return SCL_ERR_INVALID_ARG;
    static const char *names[4] = {"x", "y", "z", "intensity"};
    static const uint32_t offs[4] = {0, 4, 8, 16};
    for (int i = 0; i < 4; ++i) { fields_out[i].name = names[i]; fields_out[i].name_len = (uint32_t)std::strlen(names[i]); fields_out[i].offset = offs[i]; fields_out[i].datatype = 7 /* FLOAT32 */; fields_out[i].count = 1; }
    std::memset(cloud, 0, sizeof *cloud);
    cloud->height = 1; cloud->width = n_points;
    cloud->fields = fields_out; cloud->n_fields = 4;
    cloud->is_bigendian = 0; cloud->point_step = 32; cloud->row_step = 32u * n_points;
    cloud->data = static_cast<const uint8_t *>(points); cloud->n_data = 32u * n_points;
    cloud->is_dense = 1;
    return SCL_OK;
}

int scl_msg_cloud_xyz_layout(const scl_msg_cloud *cloud, int *stride_bytes, int *xyz_offset)
{   // the cloud was decoded from a peer's message: nothing in it is trusted
    if (!cloud || !stride_bytes || !xyz_offset) return SCL_ERR_INVALID_ARG;
    if (cloud->n_fields && !cloud->fields) return SCL_ERR_INVALID_ARG;
    int64_t off[3] = {-1, -1, -1};
    for (uint32_t i = 0; i < cloud->n_fields; ++i) {
        const scl_msg_point_field &f = cloud->fields[i];
        if (f.name_len != 1 || !f.name) continue;
        const int a = f.name[0] == 'x' ? 0 : f.name[0] == 'y' ? 1 : f.name[0] == 'z' ? 2 : -1;
        if (a < 0) continue;
        if (off[a] >= 0) return SCL_ERR_UNSUPPORTED;                              // a second x / y / z field: which one is meant?
        if (f.datatype != 7 /* FLOAT32 */ || f.count != 1) return SCL_ERR_UNSUPPORTED;
        off[a] = (int64_t)f.offset;
    }
    const uint64_t step = cloud->point_step, n = (uint64_t)cloud->width * (uint64_t)cloud->height;
    if (off[0] < 0 || off[1] != off[0] + 4 || off[2] != off[0] + 8 || cloud->is_bigendian || step < 12 || (step & 3) || (off[0] & 3) ||
        step > 0x7fffffffu || (uint64_t)off[0] + 12 > step)                      // z must end inside the record
        return SCL_ERR_UNSUPPORTED;
    // rows: pcl::toROSMsg writes row_step = point_step * width; padded rows are not the engine's (pointer, count, stride) layout
    if (cloud->row_step != 0 && (uint64_t)cloud->row_step != step * (uint64_t)cloud->width) return SCL_ERR_UNSUPPORTED;
    if (n > 0x7fffffffu || step * n > (uint64_t)cloud->n_data) return SCL_ERR_INVALID_ARG;
    // The engine's entry points read n * stride bytes from the pointer they are given (data + xyz_offset): with x not at the
    // start of the record that runs xyz_offset bytes past the last record, so the buffer must hold them
    if (n && (uint64_t)off[0] + step * n > (uint64_t)cloud->n_data) return SCL_ERR_UNSUPPORTED;
    *stride_bytes = (int)step; *xyz_offset = (int)off[0];
    return SCL_OK;
}

int scl_msg_transform_from_pose(double x, double y, double z, double roll, double pitch, double yaw, scl_msg_transform *t)
{
    if (!t) return SCL_ERR_INVALID_ARG;
    // tf::createQuaternionMsgFromRollPitchYaw (DM.h:1017): q = Rz(yaw) * Ry(pitch) * Rx(roll)
    const double cr = std::cos(0.5 * roll), sr = std::sin(0.5 * roll), cp = std::cos(0.5 * pitch), sp = std::sin(0.5 * pitch);
    const double cy = std::cos(0.5 * yaw), sy = std::sin(0.5 * yaw);
    t->translation.x = x; t->translation.y = y; t->translation.z = z;
    t->rotation.w = cr * cp * cy + sr * sp * sy;
    t->rotation.x = sr * cp * cy - cr * sp * sy;
    t->rotation.y = cr * sp * cy + sr * cp * sy;
    t->rotation.z = cr * cp * sy - sr * sp * cy;
    return SCL_OK;
}

int scl_msg_transform_to_pose(const scl_msg_transform *t, double *x, double *y, double *z, double *roll, double *pitch, double *yaw)
{
    if (!t || !x || !y || !z || !roll || !pitch || !yaw) return SCL_ERR_INVALID_ARG;
    const double qx = t->rotation.x, qy = t->rotation.y, qz = t->rotation.z, qw = t->rotation.w;
    const double n = qx * qx + qy * qy + qz * qz + qw * qw;
    if (!(n > 0.0)) return SCL_ERR_INVALID_ARG;
    const double s = 2.0 / n;
    // rotation matrix entries used by pcl::getTranslationAndEulerAngles (DM.h:1133): roll = atan2(R21, R22),
    // pitch = asin(-R20), yaw = atan2(R10, R00)
    const double r00 = 1.0 - s * (qy * qy + qz * qz), r10 = s * (qx * qy + qw * qz), r20 = s * (qx * qz - qw * qy);
    const double r21 = s * (qy * qz + qw * qx), r22 = 1.0 - s * (qx * qx + qy * qy);
    *x = t->translation.x; *y = t->translation.y; *z = t->translation.z;
    *roll = std::atan2(r21, r22);
    double sp = -r20; sp = sp > 1.0 ? 1.0 : (sp < -1.0 ? -1.0 : sp);
    *pitch = std::asin(sp);
    *yaw = std::atan2(r10, r00);
    return SCL_OK;
}

}  // extern "C"
