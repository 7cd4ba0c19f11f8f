// cu_probe.hip -- which compute unit does a workgroup run on?  (tools/cu_mask_probe.py: do CU-masked streams confine eager
// launches and graph replays?)  Every workgroup records HW_ID and XCC_ID and then spins so that the grid spreads over the chip.
#include <hip/hip_runtime.h>
extern "C" __global__ void cu_probe_kernel(unsigned* out, long long spin) {
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2 + 0] = hwid;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
extern "C" int cu_probe_launch(unsigned* out, int grid, int block, int lds_bytes, long long spin, void* stream) {
    cu_probe_kernel<<<grid, block, lds_bytes, (hipStream_t)stream>>>(out, spin);
    return (int)hipGetLastError();
}
