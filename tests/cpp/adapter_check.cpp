// Drives the engine exactly the way distributedMapping.h drives scanDescriptor
// (makeDescriptors DM.h:988-1025, globalDescriptorHandler DM.h:625-628, performIntraLoopClosure
// DM.h:1072-1086) through the C++ adapter.  Prints one line per check; exit code 0 = all good.
// Expected values come from a file written by the Python test (CPU checker results).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>

#include "pcl_types_for_adapter_check.h"
#include "scl/scan_context_hip_descriptor.hpp"
#include "scl/lidar_iris_hip_descriptor.hpp"

int main(int argc, char **argv)
{
    const int n_keyframes = argc > 1 ? std::atoi(argv[1]) : 260;
    // argv[2] = number of shards: 0 = the one-GPU constructor, G > 0 = the sharded constructor with G shards, all on
    // device 0 (a one-GPU box; on a node the list would be {0, 1, ..., 7})
    const int shards = argc > 2 ? std::atoi(argv[2]) : 0;
    auto make = [&]() -> scan_context_hip_descriptor * {
        if (shards > 0) return new scan_context_hip_descriptor(std::vector<int>((size_t)shards, 0));
        return new scan_context_hip_descriptor();
    };
    scan_context_hip_descriptor *local_impl = make();
    std::unique_ptr<scan_descriptor> scanDescriptor(local_impl);   // the DM.h:404 line
    if (scanDescriptor->getSize() != 0) { std::printf("FAIL size0\n"); return 1; }

    // a closed loop: keyframes 0..129 walk a path, 130..259 revisit it -> loops must be found
    std::mt19937_64 rng(7);
    auto uni = [&](double a, double b) { return a + (b - a) * ((rng() >> 11) * (1.0 / 9007199254740992.0)); };
    std::vector<std::array<float, 4>> boxes;
    for (int b = 0; b < 300; ++b) boxes.push_back({(float)uni(-200, 200), (float)uni(-200, 200), (float)uni(2, 8), (float)uni(1, 10)});
    std::vector<std::vector<float>> published;
    std::vector<pcl::PointCloud<pcl::PointXYZI>> all_clouds;
    for (int kf = 0; kf < n_keyframes; ++kf) {
        const int pos = kf % (n_keyframes / 2);
        const float cx = 150.0f * std::cos(0.048f * pos), cy = 150.0f * std::sin(0.048f * pos);
        pcl::PointCloud<pcl::PointXYZI> cloud;
        for (int i = 0; i < 6000; ++i) {
            const float ang = (float)uni(0, 6.283185307), rad = (float)(80.0 * std::sqrt(uni(0.001, 1)));
            pcl::PointXYZI p{};
            p.x = rad * std::cos(ang); p.y = rad * std::sin(ang); p.z = -1.65f;
            for (const auto &bx : boxes)
                if (std::fabs(p.x + cx - bx[0]) < bx[2] && std::fabs(p.y + cy - bx[1]) < bx[2]) { p.z = -1.65f + bx[3]; break; }
            p.intensity = 1.0f;
            cloud.points.push_back(p);
        }
        all_clouds.push_back(cloud);
        std::vector<float> v = scanDescriptor->makeAndSaveDescriptorAndKey(cloud, 0, kf);
        if ((int)v.size() != 20 * 60) { std::printf("FAIL vT size\n"); return 1; }
        published.push_back(v);
    }
    if (scanDescriptor->getSize() != n_keyframes) { std::printf("FAIL size\n"); return 1; }
    if (scanDescriptor->getIndex(17) != std::pair<int8_t, int>(0, 17)) { std::printf("FAIL getIndex\n"); return 1; }

    int loops = 0, correct = 0;
    unsigned long long digest = 1469598103934665603ull;    // every detection folded in: sharded and unsharded runs must print the same
    for (int cur = 0; cur < n_keyframes; ++cur) {
        const std::pair<int, float> r = scanDescriptor->detectIntraLoopClosureID(cur);
        digest = (digest ^ (unsigned long long)(unsigned)r.first) * 1099511628211ull;
        digest = (digest ^ (unsigned long long)(int)r.second) * 1099511628211ull;
        if (r.first >= 0) {
            ++loops;
            if (std::abs(r.first - (cur - n_keyframes / 2)) <= 2) ++correct;
        }
        if (cur < 104 && r.first != -1) { std::printf("FAIL early-out at %d\n", cur); return 1; }
    }
    std::printf("loops found %d, at the revisited place %d\n", loops, correct);
    if (loops < 20 || correct * 10 < loops * 8) { std::printf("FAIL loop recall\n"); return 1; }

    // a second robot ingests the same descriptors from the wire (DM.h:625-628) and must agree
    scan_context_hip_descriptor *remote_impl = make();
    std::unique_ptr<scan_descriptor> remote(remote_impl);
    for (int kf = 0; kf < n_keyframes; ++kf) remote->saveDescriptorAndKey(published[kf].data(), 1, kf);
    for (int cur = n_keyframes - 30; cur < n_keyframes; ++cur) {
        if (remote->detectIntraLoopClosureID(cur) != scanDescriptor->detectIntraLoopClosureID(cur)) { std::printf("FAIL wire parity\n"); return 1; }
    }
    // the keyframes that arrive together (round 5): the batch call returns what the virtual returned keyframe by keyframe, and the
    // one-call pipeline (descriptor + append + full-database detection per keyframe) finds the revisits
    {
        scan_context_hip_descriptor *batch_impl = make(), *pipe_impl = make();
        std::unique_ptr<scan_descriptor> batch(batch_impl), pipe(pipe_impl);
        std::vector<const pcl::PointCloud<pcl::PointXYZI> *> scans;
        std::vector<int8_t> robots((size_t)n_keyframes, 0); std::vector<int> indexs;
        for (int kf = 0; kf < n_keyframes; ++kf) { scans.push_back(&all_clouds[(size_t)kf]); indexs.push_back(kf); }
        const std::vector<std::vector<float>> vals = batch_impl->makeAndSaveDescriptorsAndKeys(scans, robots, indexs);
        if (vals != published || batch->getSize() != n_keyframes) { std::printf("FAIL batch descriptors\n"); return 1; }
        for (int cur = n_keyframes - 20; cur < n_keyframes; ++cur)
            if (batch->detectIntraLoopClosureID(cur) != scanDescriptor->detectIntraLoopClosureID(cur)) { std::printf("FAIL batch database\n"); return 1; }
        const std::vector<std::pair<int, float>> found = pipe_impl->makeSaveAndDetect(scans, robots, indexs);
        int ploops = 0, pcorrect = 0;
        for (int cur = 0; cur < n_keyframes; ++cur)
            if (found[(size_t)cur].first >= 0) { ++ploops; if (std::abs(found[(size_t)cur].first - (cur - n_keyframes / 2)) <= 2) ++pcorrect; }
        std::printf("one-call pipeline: loops found %d, at the revisited place %d\n", ploops, pcorrect);
        if (pipe->getSize() != n_keyframes || ploops < 20 || pcorrect * 10 < ploops * 8) { std::printf("FAIL pipeline loops\n"); return 1; }
    }
    // makeDescriptors in one call (filter + descriptor on the device) == scl_voxel_grid followed by the virtual
    {
        scan_context_hip_descriptor a, b;
        pcl::PointCloud<pcl::PointXYZI> raw;
        for (int i = 0; i < 30000; ++i) {
            pcl::PointXYZI p;
            p.x = (float)uni(-60, 60); p.y = (float)uni(-60, 60); p.z = (float)uni(-1.5, 6.0); p.intensity = (float)uni(0, 1);
            raw.points.push_back(p);
        }
        pcl::PointCloud<pcl::PointXYZI> filtered;
        filtered.points.resize(raw.points.size());
        int m = 0;
        if (scl_voxel_grid(a.engine(), raw.points.data(), (int)raw.points.size(), (int)sizeof(pcl::PointXYZI), 0.4f,
                           filtered.points.data(), (int)filtered.points.size(), &m) != SCL_OK) { std::printf("FAIL voxel\n"); return 1; }
        filtered.points.resize((size_t)m);
        const std::vector<float> two_calls = a.makeAndSaveDescriptorAndKey(filtered, 0, 0);
        const std::vector<float> one_call = b.makeAndSaveDescriptorAndKeyFiltered(raw, 0.4f, 0, 0);
        if (two_calls != one_call) { std::printf("FAIL filtered descriptor\n"); return 1; }
    }
    // the second descriptor of the reference through its adapter, constructed like DM.h:408 for two robots: robot 0 walks the
    // path twice (intra-robot loops), robot 1 receives robot 0's keyframes over the wire and has three of its own at revisited
    // places (inter-robot loops both ways)
    {
        const int n_iris = 100, half = 50;
        // the same rays at every visit of a place (a polar grid in the world frame: 360 azimuths x 41 ranges away from the
        // bin edges), seen under the visit's heading; `noise` > 0 perturbs ranges and heights like a second pass would --
        // without it the row key of a revisit equals the first visit's and libnabo's self-match rule drops the candidate
        auto make_cloud = [&](int pos, int yaw_deg, float noise) {
            const float cx = 150.0f * std::cos(0.048f * pos), cy = 150.0f * std::sin(0.048f * pos);
            pcl::PointCloud<pcl::PointXYZI> cloud;
            for (int a = 0; a < 360; ++a)
                for (int j = 0; j < 41; ++j) {
                    const double az = (a + 0.25) * 3.14159265358979323846 / 180.0, azs = (a + 0.25 - yaw_deg) * 3.14159265358979323846 / 180.0;
                    const float rad = 1.45f + 1.93f * j + noise * (float)uni(0.0, 0.02);
                    const float wx = rad * (float)std::cos(az), wy = rad * (float)std::sin(az);
                    pcl::PointXYZI p{};
                    p.x = rad * (float)std::cos(azs); p.y = rad * (float)std::sin(azs); p.z = -1.65f;
                    for (const auto &bx : boxes)
                        if (std::fabs(wx + cx - bx[0]) < bx[2] && std::fabs(wy + cy - bx[1]) < bx[2]) { p.z = -1.65f + bx[3] + noise * (float)uni(0.0, 0.05); break; }
                    cloud.points.push_back(p);
                }
            return cloud;
        };
        lidar_iris_hip_descriptor *i0 = new lidar_iris_hip_descriptor(80, 360, 64, 0.32, 30, 2, 10, 4, 18, 1.6f, 0.75f, 2, 0, 0, 1);
        lidar_iris_hip_descriptor *i1 = new lidar_iris_hip_descriptor(80, 360, 64, 0.32, 30, 2, 10, 4, 18, 1.6f, 0.75f, 2, 1, 0, 1);
        std::unique_ptr<scan_descriptor> iris0(i0), iris1(i1);
        for (int kf = 0; kf < n_iris; ++kf) {
            const std::vector<float> v = iris0->makeAndSaveDescriptorAndKey(make_cloud(kf % half, kf < half ? 0 : 40, kf < half ? 0.0f : 1.0f), 0, kf);
            if ((int)v.size() != 80 * 360 + 80) { std::printf("FAIL iris vT size\n"); return 1; }
            iris1->saveDescriptorAndKey(v.data(), 0, kf);
        }
        int iloops = 0, icorrect = 0;
        for (int cur = 0; cur < n_iris; ++cur) {
            const std::pair<int, float> r = iris0->detectIntraLoopClosureID(cur);
            if (cur < 41 && r.first != -1) { std::printf("FAIL iris early-out\n"); return 1; }
            if (r.first >= 0) { ++iloops; if (std::abs(r.first - (cur - half)) <= 1 && std::fabs(r.second - 40.0f) <= 1.0f) ++icorrect; }   // heading + 40 degrees = 40 columns
        }
        std::printf("iris: intra loops %d, at the revisited place and heading %d\n", iloops, icorrect);
        if (iloops < 20 || icorrect * 10 < iloops * 8) { std::printf("FAIL iris loop recall\n"); return 1; }
        int inter_ok = 0;
        for (int k = 0; k < 3; ++k) {
            const std::vector<float> v = iris1->makeAndSaveDescriptorAndKey(make_cloud(10 + 7 * k, 69, 1.0f), 1, k);
            iris0->saveDescriptorAndKey(v.data(), 1, k);
            const std::pair<int, float> mine = iris1->detectInterLoopClosureID(n_iris + k);       // own keyframe among robot 0's
            const std::pair<int, float> theirs = iris0->detectInterLoopClosureID(n_iris + k);     // received keyframe among robot 0's own
            if (mine.first >= 0 && mine.first == theirs.first && iris1->getIndex(mine.first).first == 0 &&
                iris1->getIndex(mine.first).second % half == 10 + 7 * k) ++inter_ok;
        }
        std::printf("iris: inter-robot loops agreed on %d of 3\n", inter_ok);
        if (inter_ok < 3 || iris0->getSize() != n_iris + 3 || iris0->getSize(0) != n_iris || iris0->getSize(1) != 3 || iris1->getSize(1) != 3) { std::printf("FAIL iris inter\n"); return 1; }
        i0->close(); i1->close();
        if (iris0->getSize() != 0 || iris0->detectInterLoopClosureID(0).first != -1) { std::printf("FAIL closed iris adapter\n"); return 1; }
    }
    const std::pair<int, float> inter = remote->detectInterLoopClosureID(n_keyframes - 1);
    std::printf("inter: loop %d yaw %.4f\n", inter.first, inter.second);
    std::printf("detections digest %016llx\n", digest);
    // scan_descriptor has no virtual destructor (descriptor.h:21-36): deleting through unique_ptr<scan_descriptor> would
    // not run the adapter's destructor, so a host that wants the HBM back closes explicitly first
    local_impl->close(); remote_impl->close();
    if (scanDescriptor->getSize() != 0 || scanDescriptor->detectIntraLoopClosureID(n_keyframes - 1).first != -1) { std::printf("FAIL closed adapter\n"); return 1; }
    local_impl->close();                                   // idempotent
    std::printf("ADAPTER OK\n");
    return 0;
}
