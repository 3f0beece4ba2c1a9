/* san_check -- the CPU oracle (flow2d_oracle.c) driven once under -fsanitize=address,undefined (`make -C oracle san`;
 * tests/test_sanitizers.py): whole pipelines in the three data terms on sizes that are no multiples of anything, deep
 * pyramids, every median window.  Prints an FNV-1a digest of all flow fields; the test compares it with the digest the
 * same program built without the sanitizers prints (both are built by the `san` target).  Test infrastructure only. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "flow2d_oracle.c" /* one translation unit: the oracle's source itself is what gets instrumented */

static uint64_t fnv(const void* p, size_t n, uint64_t h)
{
    const unsigned char* q = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ q[i]) * 1099511628211ull;
    return h;
}

int main(void)
{
    static const struct { size_t w, h, levels; float scale; size_t outer, inner, median; float sigma; int constancy; } cases[] = {
        {101, 67, 4, 0.5f, 2, 3, 5, 1.5f, 0}, {96, 64, 3, 0.5f, 2, 5, 3, 0.8f, 1}, {96, 64, 3, 0.5f, 1, 2, 7, 0.f, 3},
        {53, 31, 12, 0.9f, 1, 1, 1, 0.45f, 0}, {17, 9, 6, 0.7f, 2, 2, 5, 2.5f, 0}, {64, 48, 2, 0.5f, 1, 6, 5, 1.0f, 2},
    };
    uint64_t digest = 1469598103934665603ull;
    for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c) {
        const size_t n = cases[c].w * cases[c].h;
        float *f0 = malloc(n * 4), *f1 = malloc(n * 4), *u = malloc(n * 4), *v = malloc(n * 4);
        if (!f0 || !f1 || !u || !v) return 2;
        for (size_t y = 0; y < cases[c].h; ++y)
            for (size_t x = 0; x < cases[c].w; ++x) {
                f0[y * cases[c].w + x] = 128.f + 60.f * sinf(0.1f * (float)x) * cosf(0.13f * (float)y);
                f1[y * cases[c].w + x] = 128.f + 60.f * sinf(0.1f * ((float)x - 0.8f)) * cosf(0.13f * ((float)y + 0.4f));
            }
        double t = 0;
        oracle_flow_params p;
        p.warp_levels_count = cases[c].levels, p.warp_scale_factor = cases[c].scale;
        p.outer_iterations_count = cases[c].outer, p.inner_iterations_count = cases[c].inner;
        p.equation_alpha = cases[c].constancy == 3 ? 0.0005f : 35.f, p.equation_smoothness = 0.001f, p.equation_data = 0.001f;
        p.median_radius = cases[c].median, p.gaussian_sigma = cases[c].sigma, p.data_constancy = cases[c].constancy;
        p.sor_omega = c == 4 ? 1.3f : 0.f;
        const int rc = oracle_compute_flow(f0, f1, u, v, cases[c].w, cases[c].h, &p, NULL, NULL, &t);
        if (rc != 0) {
            fprintf(stderr, "san_check: case %zu: oracle_compute_flow returned %d\n", c, rc);
            return 1;
        }
        digest = fnv(u, n * 4, fnv(v, n * 4, digest));
        free(f0), free(f1), free(u), free(v);
    }
    printf("oracle san_check digest %016llx\n", (unsigned long long)digest);
    return 0;
}
