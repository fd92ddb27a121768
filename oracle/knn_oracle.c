/*
 * oracle/knn_oracle.c -- TEST INFRASTRUCTURE ONLY.
 * Brute-force restatement of what SimpleKNN::knn returns (submodules/simple-knn/simple_knn.cu:147-221): the mean of the
 * three smallest squared distances (fp32, d.x*d.x + d.y*d.y + d.z*d.z, no FMA) from each point to the OTHER points,
 * kept in ascending order exactly like updateKBest<3> (:133-145) so the final (b0 + b1 + b2) / 3 rounds identically.
 * The reference's Morton/box machinery only prunes; it does not change the result.
 */
#include <float.h>
#include <stddef.h>

void knno_dist2(int P, const float* pts, float* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        const float* p = pts + 3 * (size_t)i;
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            const float* q = pts + 3 * (size_t)j;
            const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
            float dist = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++)
                if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
