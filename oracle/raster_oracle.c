/* CPU ORACLE (test infrastructure, never shipped in the product path).
 * Builds the scalar rasterizer restatement twice: _f32 and _f64.
 * See raster_oracle_impl.h for provenance and the parity-pinning status. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUFFIX _f32
#include "raster_oracle_impl.h"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "raster_oracle_impl.h"
#undef REAL
#undef SUFFIX

/* distCUDA2 oracle: mean of the 3 smallest squared distances to OTHER points
 * (exclusion by index).  Brute force, O(N^2).  Contract: SURVEY.md Appendix B;
 * consumer /root/reference/src/models/gaussian.py:110-114.  The extension itself
 * (gitlab.inria.fr/bkerbl/simple-knn, unpinned, setup_env.sh:7) is absent here. */
void orc_knn3_mean_dist2(int N, const float* xyz, float* out) {
    for (int i = 0; i < N; ++i) {
        float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
        float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            float dx = xyz[3 * j] - x, dy = xyz[3 * j + 1] - y, dz = xyz[3 * j + 2] - z;
            float d = dx * dx + dy * dy + dz * dz;
            if (d < b2) {
                if (d < b1) {
                    b2 = b1;
                    if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
                } else {
                    b2 = d;
                }
            }
        }
        out[i] = (b0 + b1 + b2) / 3.0f;
    }
}
