// ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (gs_icp_slam_amd/).
//
// Brute-force restatement of simple_knn._C.distCUDA2: mean squared distance from each point to its 3 nearest OTHER
// points (self excluded by index).  PARITY UNPINNED: /root/reference/submodules/simple-knn is an empty directory
// (camenduru/simple-knn, commit unpinned: /root/reference/.gitmodules:8-10); the only reference site is the import at
// scene/gaussian_model.py:20 — the function is never called on the SLAM path.  Any exact 3-NN gives the same
// result up to fp rounding, so the published definition is the specification.
#include <cfloat>
#include <cmath>
extern "C" void oracle_knn_dist2(int P, const float* pts, float* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        for (int j = 0; j < P; ++j) {
            if (j == i) continue;
            const float dx = pts[3 * j] - x, dy = pts[3 * j + 1] - y, dz = pts[3 * j + 2] - z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < b2) { if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; } else b2 = d; }
        }
        // with fewer than 4 points the missing neighbours count as 0 (upstream's best[] is FLT_MAX-initialised and
        // would overflow; we define the degenerate case instead)
        float s = 0; int n = 0;
        if (b0 < FLT_MAX) { s += b0; ++n; } if (b1 < FLT_MAX) { s += b1; ++n; } if (b2 < FLT_MAX) { s += b2; ++n; }
        out[i] = n ? s / 3.0f : 0.0f;
    }
}
