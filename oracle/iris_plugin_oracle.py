"""iris_plugin_oracle.py -- CPU restatement of the plugin layer of lidar_iris_descriptor (TEST INFRASTRUCTURE ONLY;
nothing under scl_slam_amd/ or include/ imports or executes this).

Follows reference include/descriptor.h:1026-1271: the wire decoder (1026-1044), save (1046-1060), makeAndSave (1062-1083),
detectIntraLoopClosureID (1085-1151), detectInterLoopClosureID (1153-1253), getIndex / getSize (1255-1270), on top of the
C restatement of the building blocks (oracle/iris_oracle.c through tests/oracle_iris_binding.py) and the kNN of
oracle/sc_oracle.c (sco_knn: exact search, libnabo's self-match rule behind exclude_eps).

PARITY UNPINNED against the reference's binaries.  Where this restatement knowingly departs from them:
  * compare() (D.h:964-1024) runs as the reference's does -- FFT estimate, Hamming windows of five shifts, both passes as
    match_num says -- but the estimate (logPolarFFTTemplateMatch) is OpenCV's algorithms restated, not OpenCV's binaries
    (oracle/iris_oracle.c: iriso_fft_match); shift_search=1 searches every column shift instead (iriso_hamming_all);
  * libnabo's kNN arithmetic is restated as the exact search in nanoflann's accumulation order (as for the ring keys).
"""
import numpy as np

FLT_EPSILON = float(np.finfo(np.float32).eps)


class IrisPluginOracle:
    def __init__(self, oi, ob, rows=80, cols=360, nscan=64, dist_thres=0.32, num_exclude_recent=30, match_num=2,
                 num_candidates=10, nscale=4, min_wavelength=18, mult=1.6, sigma_onf=0.75, robot_num=1, this_id=0,
                 knn_exclude_eps=FLT_EPSILON, wire_decode=0, shift_search=0):
        self.oi, self.ob = oi, ob
        self.cfg = oi.config(rows, cols, nscan, nscale, min_wavelength, mult, sigma_onf)
        self.rows, self.cols = rows, cols
        self.dist_thres, self.num_exclude_recent, self.num_candidates = dist_thres, num_exclude_recent, num_candidates
        self.robot_num, self.this_id, self.eps, self.wire_decode = robot_num, this_id, knn_exclude_eps, wire_decode
        self.match_num, self.shift_search = match_num, shift_search
        self.features = [[] for _ in range(robot_num)]          # irisFeatures, D.h:1289
        self.rowkeys = [[] for _ in range(robot_num)]           # irisFeatureRowKey, D.h:1290
        self.local2global = [[] for _ in range(robot_num)]      # D.h:1291
        self.indexs = []                                        # irisFeatureIndexs, D.h:1292

    # ---- D.h:1046-1060
    def save(self, image, rowkey, robot, index):
        T, M = self.oi.encode(self.cfg, image)
        self.features[robot].append((np.array(image, np.uint8), T, M))
        self.rowkeys[robot].append(np.array(rowkey, np.float32))
        self.local2global[robot].append(len(self.indexs))
        self.indexs.append((robot, index))

    # ---- D.h:1062-1083
    def make_and_save(self, cloud, robot, index):
        img, key = self.oi.make_image(self.cfg, cloud)
        self.save(img, key, robot, index)
        return np.concatenate([img.reshape(-1).astype(np.float32), key])

    # ---- D.h:1026-1044
    def save_from_wire(self, values, robot, index):
        v = np.asarray(values, np.float32)
        rows, cols = self.rows, self.cols
        r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
        src = v[r * cols + c] if self.wire_decode else v[r * (cols + 1) + c + 1]                 # D.h:1035
        # float -> uchar the way x86 converts (cvttss2si, low byte); out of int range / NaN -> 0
        ok = np.isfinite(src) & (src > -2147483904.0) & (src < 2147483648.0)
        img = np.where(ok, np.trunc(np.where(ok, src, 0)).astype(np.int64) & 0xFF, 0).astype(np.uint8)
        self.save(img, v[rows * cols: rows * cols + rows], robot, index)

    def get_index(self, key):
        return self.indexs[key]

    def get_size(self, robot=-1):
        return len(self.indexs) if robot == -1 else len(self.local2global[robot])

    # ---- candidates + pairwise comparison, D.h:1100-1137 / 1205-1242
    def _search(self, cur_feat, cur_key, feats, keys):
        k = self.num_candidates
        idx, d2, found = self.ob.knn(np.stack(keys), cur_key, k, self.eps)
        min_dis, min_pos, min_bias = np.float32(10000000.0), -1, 0
        for i in range(k):
            if idx[i] < 0 or idx[i] >= len(feats):
                continue
            img2, T2, M2 = feats[idx[i]]
            if self.shift_search == 1:
                d, b = self.oi.hamming_all(self.cfg, cur_feat[1], cur_feat[2], T2, M2)
            else:
                d, b, _ = self.oi.compare(self.cfg, self.match_num, cur_feat[0], cur_feat[1], cur_feat[2], img2, T2, M2)   # D.h:1126
            if d < min_dis:                                                                     # NaN never wins
                min_dis, min_pos, min_bias = np.float32(d), int(idx[i]), b
        return min_pos, float(min_dis), min_bias

    # ---- D.h:1085-1151 (cur and the result are LOCAL indices of robot this_id)
    def detect_intra(self, cur):
        me = self.this_id
        if cur < self.num_exclude_recent + self.num_candidates + 1:
            return -1, 0.0, 10000000.0
        hist = cur - self.num_exclude_recent
        pos, dis, bias = self._search(self.features[me][cur], self.rowkeys[me][cur], self.features[me][:hist], self.rowkeys[me][:hist])
        if dis < self.dist_thres:
            return pos, float(bias), dis
        return -1, 0.0, dis

    # ---- D.h:1153-1253 (cur and the result are GLOBAL keys)
    def detect_inter(self, cur):
        robot, _ = self.indexs[cur]
        local = self.local2global[robot].index(cur)
        if robot == self.this_id:
            others = [i for i in range(self.robot_num) if i != self.this_id and self.local2global[i]]
        else:
            others = [self.this_id] if self.local2global[self.this_id] else []
        feats = [f for i in others for f in self.features[i]]
        keys = [k for i in others for k in self.rowkeys[i]]
        l2g = [g for i in others for g in self.local2global[i]]
        if len(l2g) < self.num_candidates + 1:
            return -1, 0.0, 10000000.0
        pos, dis, bias = self._search(self.features[robot][local], self.rowkeys[robot][local], feats, keys)
        if dis < self.dist_thres and pos >= 0:
            return l2g[pos], float(bias), dis
        return -1, 0.0, dis
