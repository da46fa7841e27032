// scan_context_hip_descriptor.hpp -- header-only adapter that plugs the MI355X engine into
// the reference's descriptor plugin interface.
//
// Include it AFTER the reference's descriptor.h (it needs `class scan_descriptor`,
// descriptor.h:21-36, and pcl::PointCloud<pcl::PointXYZI>).  It implements the same six
// virtuals as scan_context_descriptor (descriptor.h:1304-1801) with the same constructor
// arguments and return conventions, forwarding every call to the C ABI in scl_engine.h:
//
//   distributedMapping.h:404   scanDescriptor = std::unique_ptr<scan_descriptor>(new scan_context_descriptor());
//   becomes                    scanDescriptor = std::unique_ptr<scan_descriptor>(new scan_context_hip_descriptor());
//
// The reference reports nothing but "-1 = no loop" (descriptor.h:1615,1678) and logs through
// ROS; the adapter keeps that: engine errors are written to stderr and mapped to "no loop".
//
// Multi-GPU: pass the device ordinals of the node -- the keyframe database is then sharded by keyframe index over
// them behind the same six virtuals (scl_create_sharded):
//                              scanDescriptor = std::unique_ptr<scan_descriptor>(new scan_context_hip_descriptor({0, 1, 2, 3, 4, 5, 6, 7}));
//
// Lifetime: `scan_descriptor` has NO virtual destructor (descriptor.h:21-36) and distributedMapping.h owns the object
// as std::unique_ptr<scan_descriptor> (DM.h:333), so deleting through that pointer never runs
// ~scan_context_hip_descriptor(): the engine and its HBM (the whole keyframe database) would stay allocated until the
// process ends -- which is also when the reference destroys the object, so nothing leaks at run time, but a host
// that re-creates descriptors must call close() first (or give scan_descriptor a virtual destructor, one line at
// descriptor.h:23).  close() is idempotent; every call after it reports "no engine" and returns "no loop".
#pragma once

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

#include "scl_engine.h"

class scan_context_hip_descriptor : public scan_descriptor
{
public:
    // same parameter list and defaults as scan_context_descriptor's ctor, descriptor.h:1307-1316
    scan_context_hip_descriptor(
        int numRing            = 20,
        int numSector          = 60,
        int numCandidates      = 3,
        double distThres       = 0.14,
        double lidarHeight     = 1.65,
        double maxRadius       = 80.0,
        int numExcludeRecent   = 100,
        int treeMakingPeriod   = 10,
        double searchRatio     = 0.1,
        int device             = 0,
        // The live intra-robot search is libnabo's knn with optionFlags = 0 (descriptor.h:1631-1642), which skips
        // neighbours whose squared ring-key distance is <= FLT_EPSILON (no self match): a yaw-only revisit has the
        // SAME ring key as the query, and libnabo does not return it.  0 selects nanoflann's behaviour (every key
        // counts), which the inter-robot path uses regardless of this value (descriptor.h:1710-1716).
        float knnExcludeEps    = FLT_EPSILON)
    {
        init(numRing, numSector, numCandidates, distThres, lidarHeight, maxRadius, numExcludeRecent, treeMakingPeriod,
             searchRatio, knnExcludeEps, &device, 1, false);
    }

    // the same database sharded by keyframe index over several GPUs (keyframe g on devices[g % devices.size()])
    explicit scan_context_hip_descriptor(
        const std::vector<int> &devices,
        int numRing            = 20,
        int numSector          = 60,
        int numCandidates      = 3,
        double distThres       = 0.14,
        double lidarHeight     = 1.65,
        double maxRadius       = 80.0,
        int numExcludeRecent   = 100,
        int treeMakingPeriod   = 10,
        double searchRatio     = 0.1,
        float knnExcludeEps    = FLT_EPSILON)
    {
        init(numRing, numSector, numCandidates, distThres, lidarHeight, maxRadius, numExcludeRecent, treeMakingPeriod,
             searchRatio, knnExcludeEps, devices.data(), static_cast<int>(devices.size()), true);
    }

    ~scan_context_hip_descriptor() { close(); }
    // releases the engine and its HBM; see "Lifetime" above
    void close()
    {
        if (engine_) scl_destroy(engine_);
        engine_ = nullptr;
    }
    scan_context_hip_descriptor(const scan_context_hip_descriptor &) = delete;
    scan_context_hip_descriptor &operator=(const scan_context_hip_descriptor &) = delete;

    // descriptor.h:25 / 1604-1611; pcl::PointXYZI is a 32-byte record with x,y,z first
    std::vector<float> makeAndSaveDescriptorAndKey(const pcl::PointCloud<pcl::PointXYZI> &scan,
                                                   const int8_t robot, const int index) override
    {
        std::vector<float> vT(static_cast<size_t>(cells_), 0.0f);
        report(scl_make_and_save(engine_, scan.points.data(), static_cast<int>(scan.points.size()),
                                 static_cast<int>(sizeof(pcl::PointXYZI)), robot, index, vT.data()),
               "makeAndSaveDescriptorAndKey");
        return vT;
    }

    // descriptor.h:27 / 1572-1585 (descriptorMat = global_descriptor.values.data(), DM.h:627)
    void saveDescriptorAndKey(const float *descriptorMat, const int8_t robot, const int index) override
    {
        report(scl_save_from_wire(engine_, descriptorMat, robot, index), "saveDescriptorAndKey");
    }

    // descriptor.h:29 / 1613-1674: {loop index or -1, ring shift as float}
    std::pair<int, float> detectIntraLoopClosureID(const int currentPtr) override
    {
        int loop_id = -1; float shift = 0.0f;
        if (!report(scl_detect_intra(engine_, currentPtr, &loop_id, &shift, nullptr), "detectIntraLoopClosureID"))
            return std::pair<int, float>(-1, 0.0f);
        return std::pair<int, float>(loop_id, shift);
    }

    // descriptor.h:31 / 1676-1756: {loop index or -1, relative yaw in radians}
    std::pair<int, float> detectInterLoopClosureID(const int currentPtr) override
    {
        int loop_id = -1; float yaw = 0.0f;
        if (!report(scl_detect_inter(engine_, currentPtr, &loop_id, &yaw, nullptr), "detectInterLoopClosureID"))
            return std::pair<int, float>(-1, 0.0f);
        return std::pair<int, float>(loop_id, yaw);
    }

    // descriptor.h:33 / 1758-1761
    std::pair<int8_t, int> getIndex(const int key) override
    {
        int8_t robot = 0; int index = -1;
        report(scl_get_index(engine_, key, &robot, &index), "getIndex");
        return std::pair<int8_t, int>(robot, index);
    }

    // descriptor.h:35 / 1763-1766
    int getSize(const int idIn = -1) override
    {
        const int n = scl_get_size(engine_, idIn);
        return n < 0 ? 0 : n;
    }

    scl_engine *engine() { return engine_; }   // for the geometric-verification calls (scl_icp_align ...)

    // Not part of scan_descriptor: makeDescriptors' filter + descriptor + append (DM.h:996-1002) in one call; the
    // filtered cloud never leaves the device.  Same return value as makeAndSaveDescriptorAndKey on the
    // VoxelGrid(leaf)-filtered scan.
    std::vector<float> makeAndSaveDescriptorAndKeyFiltered(const pcl::PointCloud<pcl::PointXYZI> &rawScan, float leaf,
                                                           const int8_t robot, const int index)
    {
        std::vector<float> vT(static_cast<size_t>(cells_), 0.0f);
        report(scl_make_and_save_filtered(engine_, rawScan.points.data(), static_cast<int>(rawScan.points.size()),
                                          static_cast<int>(sizeof(pcl::PointXYZI)), leaf, robot, index, vT.data(), nullptr),
               "makeAndSaveDescriptorAndKeyFiltered");
        return vT;
    }

    // Not part of scan_descriptor: makeAndSaveDescriptorAndKey for the keyframes that arrive together (DM.h:988-1025 once per keyframe),
    // in order -- groups of 16 cost two kernel launches each, the next group's clouds travel meanwhile (pinned buffers: scl_host_alloc /
    // scl_host_register).  values[i] = what makeAndSaveDescriptorAndKey(*scans[i], robots[i], indexs[i]) returns.
    std::vector<std::vector<float>> makeAndSaveDescriptorsAndKeys(const std::vector<const pcl::PointCloud<pcl::PointXYZI> *> &scans,
                                                                  const std::vector<int8_t> &robots, const std::vector<int> &indexs)
    {
        const size_t n = scans.size();
        std::vector<const void *> clouds(n); std::vector<int> counts(n);
        for (size_t i = 0; i < n; ++i) { clouds[i] = scans[i]->points.data(); counts[i] = static_cast<int>(scans[i]->points.size()); }
        std::vector<float> flat(n * static_cast<size_t>(cells_), 0.0f);
        std::vector<std::vector<float>> out;
        if (robots.size() != n || indexs.size() != n ||
            !report(scl_make_and_save_many(engine_, clouds.data(), counts.data(), static_cast<int>(n), static_cast<int>(sizeof(pcl::PointXYZI)),
                                           robots.data(), indexs.data(), flat.data()), "makeAndSaveDescriptorsAndKeys"))
            return out;
        for (size_t i = 0; i < n; ++i) out.emplace_back(flat.begin() + static_cast<long>(i * cells_), flat.begin() + static_cast<long>((i + 1) * cells_));
        return out;
    }

    // ... and, in the same call, the full-database detection of every new keyframe over [0, key - NUM_EXCLUDE_RECENT) (descriptor.h:1627):
    // loops[i] = {keyframe index or -1 (threshold SC_DIST_THRES applied, descriptor.h:1662), ring shift as float} like detectIntraLoopClosureID
    std::vector<std::pair<int, float>> makeSaveAndDetect(const std::vector<const pcl::PointCloud<pcl::PointXYZI> *> &scans,
                                                         const std::vector<int8_t> &robots, const std::vector<int> &indexs, double distThres = 0.14)
    {
        const size_t n = scans.size();
        std::vector<const void *> clouds(n); std::vector<int> counts(n), nn(n, -1), shift(n, 0);
        std::vector<double> dist(n, 1e7);
        for (size_t i = 0; i < n; ++i) { clouds[i] = scans[i]->points.data(); counts[i] = static_cast<int>(scans[i]->points.size()); }
        std::vector<std::pair<int, float>> out(n, std::pair<int, float>(-1, 0.0f));
        if (robots.size() != n || indexs.size() != n ||
            !report(scl_stream_from_points(engine_, clouds.data(), counts.data(), static_cast<int>(n), static_cast<int>(sizeof(pcl::PointXYZI)),
                                           robots.data(), indexs.data(), nn.data(), shift.data(), dist.data(), nullptr), "makeSaveAndDetect"))
            return out;
        for (size_t i = 0; i < n; ++i)
            if (nn[i] >= 0 && dist[i] < distThres) out[i] = std::pair<int, float>(nn[i], static_cast<float>(shift[i]));
        return out;
    }

private:
    void init(int numRing, int numSector, int numCandidates, double distThres, double lidarHeight, double maxRadius,
              int numExcludeRecent, int treeMakingPeriod, double searchRatio, float knnExcludeEps,
              const int *devices, int n_devices, bool sharded)
    {
        scl_config cfg;
        scl_default_config(&cfg);
        cfg.num_ring = numRing;                 cfg.num_sector = numSector;
        cfg.num_candidates = numCandidates;     cfg.dist_thres = distThres;
        cfg.lidar_height = lidarHeight;         cfg.max_radius = maxRadius;
        cfg.num_exclude_recent = numExcludeRecent;
        cfg.tree_making_period = treeMakingPeriod;
        cfg.search_ratio = searchRatio;         cfg.device = n_devices > 0 ? devices[0] : 0;
        cfg.knn_exclude_eps = knnExcludeEps;
        cells_ = numRing * numSector;
        const int rc = sharded ? scl_create_sharded(&cfg, devices, n_devices, 0, &engine_) : scl_create(&cfg, &engine_);
        if (rc != SCL_OK) {
            std::fprintf(stderr, "[scan_context_hip_descriptor] engine creation failed: %s\n", scl_status_string(rc));
            engine_ = nullptr;
        }
    }

    bool report(int rc, const char *where) const
    {
        if (!engine_) {
            std::fprintf(stderr, "[scan_context_hip_descriptor] %s: no engine (creation failed or close() was called)\n", where);
            return false;
        }
        if (rc == SCL_OK) return true;
        std::fprintf(stderr, "[scan_context_hip_descriptor] %s: %s (%s)\n", where, scl_status_string(rc),
                     engine_ ? scl_last_error(engine_) : "no engine");
        return false;
    }

    scl_engine *engine_ = nullptr;
    int cells_ = 0;
};
