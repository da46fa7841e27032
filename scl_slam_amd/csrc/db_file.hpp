// db_file.hpp -- the database dump's file format and its PARSER, host code only (no HIP): scl_db_load_file (engine.hip) hands the
// parsed chunks to scl_save_bulk; tests/cpp/fuzz_host.cpp runs the same parser over mutated bytes under ASan / UBSan (make sanitize).
//   header | float32[count][R*S] descriptors in wire order (D.h:1446-1455) | int32[count][2] (robot, index) (D.h:1758-1761)
// Nothing is sized by the header before the file has proved to be exactly as long as the header says.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

namespace scl {

struct DbFileHeader {
    char magic[8];                                          // "SCLDB\0\0\1"
    int32_t version, num_ring, num_sector, count;
    int32_t reserved[4];
};
static const char kDbMagic[8] = {'S', 'C', 'L', 'D', 'B', 0, 0, 1};

enum DbFileStatus { DBF_OK = 0, DBF_NOT_A_DUMP, DBF_OTHER_GRID, DBF_CORRUPT_HEADER, DBF_LENGTH, DBF_TRUNCATED, DBF_NOMEM, DBF_SINK };

inline const char *db_file_status_string(int s)
{
    switch (s) {
    case DBF_OK: return "ok";
    case DBF_NOT_A_DUMP: return "db_load: not a database dump of this engine";
    case DBF_OTHER_GRID: return "db_load: the dump was made for another grid (rings x sectors)";
    case DBF_CORRUPT_HEADER: return "db_load: corrupt header";
    case DBF_LENGTH: return "db_load: file length does not match the keyframe count of its header";
    case DBF_TRUNCATED: return "db_load: truncated file";
    case DBF_NOMEM: return "db_load: out of host memory";
    default: return "db_load: the engine refused a chunk";
    }
}

// Reads a dump from `f` (seekable) for an R x S engine and hands it to `sink(values, count, robots, indexs)` in chunks of up to 512
// descriptors (a non-zero return stops the parse: DBF_SINK, *sink_rc = that value).  *header receives the header when it is valid.
template <class Sink>
int db_file_parse(FILE *f, int R, int S, DbFileHeader *header, Sink &&sink, int *sink_rc)
{
    DbFileHeader h;
    memset(&h, 0, sizeof h);
    if (sink_rc) *sink_rc = 0;
    if (R <= 0 || S <= 0) return DBF_OTHER_GRID;
    const size_t cells = (size_t)R * (size_t)S;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kDbMagic, 8) != 0 || h.version != 1) return DBF_NOT_A_DUMP;
    if (h.num_ring != R || h.num_sector != S) return DBF_OTHER_GRID;
    if (h.count < 0) return DBF_CORRUPT_HEADER;
    const unsigned long long want = (unsigned long long)sizeof h + (unsigned long long)h.count * (sizeof(float) * cells + 2 * sizeof(int32_t));
    long long have = -1;
    if (fseeko(f, 0, SEEK_END) == 0) have = (long long)ftello(f);
    if (have < 0 || (unsigned long long)have != want) return DBF_LENGTH;
    if (header) *header = h;
    try {
        // the index map sits behind the descriptors: read it first, then stream the descriptors in chunks
        std::vector<int8_t> robots((size_t)h.count);
        std::vector<int> indexs((size_t)h.count);
        if (fseeko(f, (off_t)(sizeof h + sizeof(float) * cells * (size_t)h.count), SEEK_SET) != 0) return DBF_TRUNCATED;
        for (int k = 0; k < h.count; ++k) {
            int32_t rec[2];
            if (fread(rec, sizeof rec, 1, f) != 1) return DBF_TRUNCATED;
            robots[(size_t)k] = (int8_t)rec[0]; indexs[(size_t)k] = rec[1];
        }
        if (fseeko(f, (off_t)sizeof h, SEEK_SET) != 0) return DBF_TRUNCATED;
        const int chunk = 512;
        std::vector<float> buf(cells * (size_t)(h.count < chunk ? h.count : chunk));
        for (int done = 0; done < h.count; done += chunk) {
            const int c = h.count - done < chunk ? h.count - done : chunk;
            if (fread(buf.data(), sizeof(float) * cells, (size_t)c, f) != (size_t)c) return DBF_TRUNCATED;
            const int rc = sink(buf.data(), c, robots.data() + done, indexs.data() + done);
            if (rc) { if (sink_rc) *sink_rc = rc; return DBF_SINK; }
        }
    } catch (const std::bad_alloc &) {
        return DBF_NOMEM;
    }
    return DBF_OK;
}

}  // namespace scl
