// fuzz_host.cpp -- TEST INFRASTRUCTURE: libFuzzer harness over the engine's HOST-ONLY parsers of untrusted bytes, built with
// -fsanitize=fuzzer,address,undefined by `make sanitize` (no GPU, no HIP):
//   * the ROS 1 wire decoders of the peers' messages (scl_slam_amd/csrc/messages.hip: global_descriptor DM.h:556-629, loop_info,
//     geometric_verification request / response DM.h:1189-1268) -- a decoded message's strings, values and cloud bytes point INTO
//     the input, so every byte they claim is read here (ASan sees a claim past the buffer); what decodes is re-encoded and must
//     decode again to the same lengths; the cloud layout check (scl_msg_cloud_xyz_layout) runs on every decoded cloud and a cloud it
//     accepts is walked record by record the way the engine's (pointer, count, stride) entry points would;
//   * the database dump parser (scl_slam_amd/csrc/db_file.hpp, behind scl_db_load_file) through fmemopen.
// First input byte: which target (mod 5); the rest: the bytes.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "db_file.hpp"
#include "scl_engine.h"
#include "scl_messages.h"

// the codec itself, in this translation unit (host C++: the file has no device code)
#include "messages.hip"

static volatile uint64_t g_sink;

static void touch(const void *p, size_t n)
{
    const uint8_t *b = static_cast<const uint8_t *>(p);
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) s += b[i];
    g_sink += s;
}

static void touch_header(const scl_msg_header &h) { if (h.frame_id) touch(h.frame_id, h.frame_id_len); }

static void check_cloud(const scl_msg_cloud &c)
{
    touch_header(c.header);
    for (uint32_t i = 0; i < c.n_fields && i < 16; ++i) if (c.fields[i].name) touch(c.fields[i].name, c.fields[i].name_len);
    if (c.data) touch(c.data, c.n_data);
    int stride = 0, off = 0;
    if (scl_msg_cloud_xyz_layout(&c, &stride, &off) == SCL_OK) {
        // accepted: width * height records of `stride` bytes from data + off, 12 bytes of x, y, z each, must lie inside data
        const uint64_t n = (uint64_t)c.width * c.height;
        for (uint64_t i = 0; i < n; ++i) touch(c.data + off + i * (uint64_t)stride, 12);
    }
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size)
{
    if (size < 1) return 0;
    const int target = data[0] % 5;
    const uint8_t *p = data + 1;
    const size_t n = size - 1;
    std::vector<uint8_t> out;
    size_t len = 0;
    switch (target) {
    case 0: {
        scl_msg_global_descriptor m;
        if (scl_msg_global_descriptor_decode(p, n, &m) == SCL_OK) {
            touch_header(m.header);
            if (m.values) touch(m.values, sizeof(float) * (size_t)m.n_values);
            scl_msg_global_descriptor_encode(&m, nullptr, 0, &len);
            out.resize(len);
            if (scl_msg_global_descriptor_encode(&m, out.data(), out.size(), &len) != SCL_OK || len != n) __builtin_trap();
            scl_msg_global_descriptor m2;
            if (scl_msg_global_descriptor_decode(out.data(), len, &m2) != SCL_OK || m2.n_values != m.n_values) __builtin_trap();
        }
        break;
    }
    case 1: {
        scl_msg_loop_info m;
        if (scl_msg_loop_info_decode(p, n, &m) == SCL_OK) {
            touch_header(m.header);
            scl_msg_loop_info_encode(&m, nullptr, 0, &len);
            out.resize(len);
            if (scl_msg_loop_info_encode(&m, out.data(), out.size(), &len) != SCL_OK || len != n) __builtin_trap();
        }
        break;
    }
    case 2: {
        scl_msg_geometric_verification_request m;
        if (scl_msg_geometric_verification_request_decode(p, n, &m) == SCL_OK) {
            check_cloud(m.featureCloud);
            scl_msg_geometric_verification_request_encode(&m, nullptr, 0, &len);
            out.resize(len);
            if (scl_msg_geometric_verification_request_encode(&m, out.data(), out.size(), &len) != SCL_OK || len != n) __builtin_trap();
        }
        break;
    }
    case 3: {
        scl_msg_geometric_verification_response m;
        if (scl_msg_geometric_verification_response_decode(p, n, &m) == SCL_OK) {
            scl_msg_geometric_verification_response_encode(&m, nullptr, 0, &len);
            out.resize(len);
            if (scl_msg_geometric_verification_response_encode(&m, out.data(), out.size(), &len) != SCL_OK || len != n) __builtin_trap();
        }
        break;
    }
    default: {
        if (n == 0) break;
        FILE *f = fmemopen(const_cast<uint8_t *>(p), n, "rb");
        if (!f) break;
        scl::DbFileHeader h;
        int sink_rc = 0;
        uint64_t kf = 0;
        // a tiny grid, so that the fuzzer can reach the descriptor and index sections (4 x 6 cells = 96 bytes per keyframe)
        const int st = scl::db_file_parse(f, 4, 6, &h, [&](const float *vals, int c, const int8_t *robots, const int *indexs) {
            touch(vals, sizeof(float) * 24 * (size_t)c); touch(robots, (size_t)c); touch(indexs, sizeof(int) * (size_t)c);
            kf += (uint64_t)c;
            return 0;
        }, &sink_rc);
        if (st == scl::DBF_OK && kf != (uint64_t)h.count) __builtin_trap();
        fclose(f);
        break;
    }
    }
    return 0;
}
