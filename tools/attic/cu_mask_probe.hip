// Probe hipExtStreamCreateWithCUMask on gfx950: which XCDs / CUs do the workgroups of a masked stream land on?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(512) void where(unsigned* out) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;         // HW_REG_XCC_ID[3:0]
        const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);               // HW_REG_HW_ID (cu, sh, se fields)
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hwid;
    }
    // keep the CU busy a little so that workgroups spread
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}
static void run(const char* name, const std::vector<uint32_t>& mask, int nwg) {
    hipStream_t st;
    hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: stream create failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned* d; hipMalloc(&d, nwg * 8); hipMemset(d, 0xFF, nwg * 8);
    where<<<nwg, 512, 0, st>>>(d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nwg * 2); hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < nwg; ++i) cnt[h[2 * i] & 15]++;
    printf("%-34s %3d workgroups -> per XCD:", name, nwg);
    for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
    printf("\n");
    hipFree(d); hipStreamDestroy(st);
}
int main() {
    run("no mask", {}, 256);
    run("bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}, 256);
    run("bits 0..63", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}, 256);
    run("bits 192..255", {0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu}, 256);
    run("every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}, 256);
    run("bits 0..7 of every 32", {0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu}, 256);
    return 0;
}
