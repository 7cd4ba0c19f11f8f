/*
 * ref_shim.hip -- C-ABI doorway onto the REFERENCE's own device code (test infrastructure).
 *
 * oracle/_ref/libpn2_ref_{off,fast}.so = the reference's tf_ops/tf_sampling.cu and
 * tf_ops/tf_grouping.cu, compiled UNMODIFIED from where they lie under /root/reference
 * (oracle/Makefile target `_ref`; both files contain no #include, so
 * `hipcc -x hip -include hip/hip_runtime.h` is all they need), plus this file, which only
 * forwards extern "C" entry points to the reference's C++-linkage launchers
 * (tf_sampling.cu:208-229, tf_grouping.cu:138-162) and supplies what the TensorFlow op glue
 * supplies around them: the zero fill of gradient outputs (tf_sampling.cpp:236,
 * tf_grouping.cpp:271) and a device synchronise.  Nothing of the reference is copied here.
 *
 * The launchers use the null stream (`<<<grid, block>>>`), so every wrapper synchronises
 * the device before and after: callers (tests) may have work queued on torch's stream.
 *
 * Two builds: `-ffp-contract=off` (every mul/add rounded) and `-ffp-contract=fast`
 * (LLVM free to contract, the analogue of nvcc's default --fmad=true).  The tests compare
 * both against the C restatement's arithmetic modes and against the HIP kernels.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <hip/hip_runtime.h>

// C++-linkage launchers defined by the reference translation units.
void farthestpointsamplingLauncher(int b, int n, int m, const float* inp, float* temp, int* out);
void gatherpointLauncher(int b, int n, int m, const float* inp, const int* idx, float* out);
void scatteraddpointLauncher(int b, int n, int m, const float* out_g, const int* idx, float* inp_g);
void probsampleLauncher(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out);
void queryBallPointLauncher(int b, int n, int m, float radius, int nsample, const float* xyz1,
                            const float* xyz2, int* idx, int* pts_cnt);
void selectionSortLauncher(int b, int n, int m, int k, const float* dist, int* outi, float* out);
void groupPointLauncher(int b, int n, int c, int m, int nsample, const float* points, const int* idx,
                        float* out);
void groupPointGradLauncher(int b, int n, int c, int m, int nsample, const float* grad_out,
                            const int* idx, float* grad_points);

static int finish() {
    hipError_t e = hipGetLastError();
    hipError_t s = hipDeviceSynchronize();
    return (int)(e != hipSuccess ? e : s);
}

extern "C" {

// temp: (32, n) floats, as tf_sampling.cpp:146-148 allocates it.
int ref_farthest_point_sample(int b, int n, int m, const float* inp, float* temp, int* out) {
    hipDeviceSynchronize();
    farthestpointsamplingLauncher(b, n, m, inp, temp, out);
    return finish();
}

int ref_gather_point(int b, int n, int m, const float* inp, const int* idx, float* out) {
    hipDeviceSynchronize();
    gatherpointLauncher(b, n, m, inp, idx, out);
    return finish();
}

int ref_gather_point_grad(int b, int n, int m, const float* out_g, const int* idx, float* inp_g) {
    hipDeviceSynchronize();
    hipMemset(inp_g, 0, sizeof(float) * (size_t)b * n * 3);  // tf_sampling.cpp:236
    scatteraddpointLauncher(b, n, m, out_g, idx, inp_g);
    return finish();
}

// temp: (b, n) floats (tf_sampling.cpp:100-102).
int ref_prob_sample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out) {
    hipDeviceSynchronize();
    probsampleLauncher(b, n, m, inp_p, inp_r, temp, out);
    return finish();
}

int ref_query_ball_point(int b, int n, int m, float radius, int nsample, const float* xyz1,
                         const float* xyz2, int* idx, int* pts_cnt) {
    hipDeviceSynchronize();
    queryBallPointLauncher(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt);
    return finish();
}

int ref_selection_sort(int b, int n, int m, int k, const float* dist, int* outi, float* out) {
    hipDeviceSynchronize();
    selectionSortLauncher(b, n, m, k, dist, outi, out);
    return finish();
}

int ref_group_point(int b, int n, int c, int m, int nsample, const float* points, const int* idx,
                    float* out) {
    hipDeviceSynchronize();
    groupPointLauncher(b, n, c, m, nsample, points, idx, out);
    return finish();
}

int ref_group_point_grad(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                         float* grad_points) {
    hipDeviceSynchronize();
    hipMemset(grad_points, 0, sizeof(float) * (size_t)b * n * c);  // tf_grouping.cpp:271
    groupPointGradLauncher(b, n, c, m, nsample, grad_out, idx, grad_points);
    return finish();
}

const char* ref_build_info(void) {
#ifdef PN2_REF_CONTRACT
    return "reference tf_sampling.cu + tf_grouping.cu, hipcc gfx950, -ffp-contract=" PN2_REF_CONTRACT;
#else
    return "reference tf_sampling.cu + tf_grouping.cu, hipcc gfx950";
#endif
}

}  // extern "C"
