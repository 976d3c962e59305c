// When does the physical memory of a HIP VMM chunk go back to the device (ROCm 7.2, gfx950)?  hipMemGetInfo after every
// step of four life cycles, and the proof by allocation (can a hipMalloc have the memory?).
//   hipcc --offload-arch=gfx950 -O2 tools/lab/vmm_meminfo.hip -o /tmp/vmm_meminfo && /tmp/vmm_meminfo [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static double free_gib() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f / 1073741824.0; }
static const size_t CHUNK = (size_t)1 << 30;
static hipMemAllocationProp prop_of() { hipMemAllocationProp p = {}; p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; p.location.id = 0; return p; }
static int can_malloc(double gib) { void* p = nullptr; const hipError_t e = hipMalloc(&p, (size_t)(gib * 1073741824.0)); if (e == hipSuccess) { (void)hipMemset(p, 3, 1 << 20); (void)hipFree(p); return 1; } (void)hipGetLastError(); return 0; }

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atoi(argv[1]) : 32;
    CK(hipSetDevice(0));
    const hipMemAllocationProp prop = prop_of();
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const double f0 = free_gib();
    printf("{\"chunks_GiB\": %zu, \"free_at_start\": %.2f", n, f0);
    {   // 1: never mapped
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (auto& x : h) CK(hipMemCreate(&x, CHUNK, &prop, 0));
        const double a = free_gib();
        for (auto& x : h) CK(hipMemRelease(x));
        printf(", \"created_only\": {\"after_create\": %.2f, \"after_release\": %.2f}", a, free_gib());
    }
    {   // 2: map, touch, unmap, release, synchronize, free the addresses
        void* va = nullptr; CK(hipMemAddressReserve(&va, n * CHUNK, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], CHUNK, &prop, 0)); CK(hipMemMap((char*)va + i * CHUNK, CHUNK, 0, h[i], 0)); }
        CK(hipMemSetAccess(va, n * CHUNK, &acc, 1));
        CK(hipMemset(va, 1, n * CHUNK)); CK(hipDeviceSynchronize());
        const double a = free_gib();
        for (size_t i = 0; i < n; ++i) CK(hipMemUnmap((char*)va + i * CHUNK, CHUNK));
        const double b = free_gib();
        for (size_t i = 0; i < n; ++i) CK(hipMemRelease(h[i]));
        const double c = free_gib();
        CK(hipDeviceSynchronize());
        const double d = free_gib();
        const int m1 = can_malloc(f0 - 4);
        CK(hipMemAddressFree(va, n * CHUNK));
        const double e = free_gib();
        const int m2 = can_malloc(f0 - 4);
        printf(", \"map_unmap_release_free\": {\"mapped\": %.2f, \"after_unmap\": %.2f, \"after_release\": %.2f, \"after_sync\": %.2f, "
               "\"hipMalloc_of_all_but_4GiB\": %d, \"after_address_free\": %.2f, \"hipMalloc_then\": %d}", a, b, c, d, m1, e, m2);
    }
    {   // 3: the CUDA idiom -- release the handle right after mapping, the mapping keeps the memory alive; unmap later
        void* va = nullptr; CK(hipMemAddressReserve(&va, n * CHUNK, 0, nullptr, 0));
        for (size_t i = 0; i < n; ++i) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, CHUNK, &prop, 0)); CK(hipMemMap((char*)va + i * CHUNK, CHUNK, 0, h, 0)); CK(hipMemRelease(h)); }
        CK(hipMemSetAccess(va, n * CHUNK, &acc, 1));
        CK(hipMemset(va, 1, n * CHUNK)); CK(hipDeviceSynchronize());
        const double a = free_gib();
        for (size_t i = 0; i < n; ++i) CK(hipMemUnmap((char*)va + i * CHUNK, CHUNK));
        const double b = free_gib();
        const int m1 = can_malloc(f0 - 4);
        CK(hipMemAddressFree(va, n * CHUNK));
        printf(", \"release_right_after_map\": {\"mapped\": %.2f, \"after_unmap\": %.2f, \"hipMalloc_of_all_but_4GiB\": %d, \"after_address_free\": %.2f}", a, b, m1, free_gib());
    }
    {   // 4: no hipMemSetAccess / no touch: map, unmap, release
        void* va = nullptr; CK(hipMemAddressReserve(&va, n * CHUNK, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], CHUNK, &prop, 0)); CK(hipMemMap((char*)va + i * CHUNK, CHUNK, 0, h[i], 0)); }
        const double a = free_gib();
        for (size_t i = 0; i < n; ++i) { CK(hipMemUnmap((char*)va + i * CHUNK, CHUNK)); CK(hipMemRelease(h[i])); }
        printf(", \"mapped_never_accessed\": {\"mapped\": %.2f, \"after_unmap_release\": %.2f}", a, free_gib());
    }
    {   // 5: quarantine -- unmap, release, free the addresses (the memory comes back) and reserve the SAME addresses again at once,
        //    nothing mapped: the range can never be handed out again, and costs no memory
        void* va = nullptr; CK(hipMemAddressReserve(&va, n * CHUNK, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], CHUNK, &prop, 0)); CK(hipMemMap((char*)va + i * CHUNK, CHUNK, 0, h[i], 0)); }
        CK(hipMemSetAccess(va, n * CHUNK, &acc, 1));
        CK(hipMemset(va, 1, n * CHUNK)); CK(hipDeviceSynchronize());
        const double a = free_gib();
        for (size_t i = 0; i < n; ++i) { CK(hipMemUnmap((char*)va + i * CHUNK, CHUNK)); CK(hipMemRelease(h[i])); }
        CK(hipMemAddressFree(va, n * CHUNK));
        void* again = nullptr; CK(hipMemAddressReserve(&again, n * CHUNK, 0, va, 0));
        const double b = free_gib();
        const int m = can_malloc(b - 4);
        void* other = nullptr; CK(hipMemAddressReserve(&other, n * CHUNK, 0, va, 0));      // a second request for the same place must get another
        printf(", \"quarantine\": {\"mapped\": %.2f, \"same_address_again\": %d, \"free_after\": %.2f, \"hipMalloc_of_all_but_4GiB\": %d, "
               "\"second_reservation_elsewhere\": %d}", a, again == va, b, m, other != va);
        // 6: partial release -- a range reserved per CHUNK at hinted consecutive addresses: can the middle be given back?
    }
    printf(", \"free_at_end\": %.2f}\n", free_gib());
    return 0;
}
