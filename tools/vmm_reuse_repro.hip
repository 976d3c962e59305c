// vmm_reuse_repro.hip -- standalone reproducer (plain HIP, no libdswx) for the hazard fenced in
// proteus_amd/csrc/dswx_batch.hip (VmRange): is a virtual address range that was unmapped and handed back
// (hipMemAddressFree), then reserved again and mapped onto NEW physical memory, safe to write through?
//   hipcc --offload-arch=gfx950 -O2 tools/vmm_reuse_repro.hip -o /tmp/vmm_repro && /tmp/vmm_repro [mode] [iters] [chunk_MiB] [chunks]
//   mode 0: unmap + release + hipMemAddressFree, then hipMemAddressReserve with the old address as hint   (what round 3 did)
//   mode 1: unmap + release, the reservation KEPT, new handles mapped at the same addresses           (a free list of ranges)
//   mode 2: mode 0 with hipDeviceSynchronize() before every unmap                                       
//   mode 4: mode 1, and between the unmap and the new mapping a kernel reads one word of every 64 KiB of a 16 GiB
//           allocation (8192 distinct 2-MiB translations): if the failures go away with it, what is stale is a cached
//           translation (a TLB entry the unmap did not invalidate) that this traffic evicts
//   mode 7: mode 1, and in that place a kernel reads a word of each of 6144 SEPARATE 2-MiB VMM chunks (one translation
//           each, unlike the large fragments of a single hipMalloc) three times over
//   mode 5: mode 1 with a 200 ms sleep in that place (the control for mode 4: time alone)
//   mode 6: mode 1 without the hipMalloc / hipFree churn around the re-mapping (is the failure tied to it?)
//   mode 3: the library's own pattern (dswx_batch_place_slide): a current range stays alive while a wider one is reserved
//           and mapped beside it, written and checked; half of the wide range's chunks are unmapped ("trim"), the
//           current range is unmapped + released + hipMemAddressFree'd, the wide one becomes current; sizes vary, so
//           a later reservation can land on addresses an earlier, freed range used
// Every iteration: map A, kernel fills A with pattern a, [free / re-reserve], map B at the same VA, kernel fills B with
// pattern b, then three readers of B: a checking kernel, hipMemcpy D2H, and (after a hipMemcpy H2D of pattern c) the
// checking kernel again.  Prints one JSON line with the number of iterations each reader saw wrong data in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v ^ (uint32_t)i, p + i);
}
__global__ void thrash(const uint32_t* p, size_t words, size_t stride_words, unsigned long long* sink) {
    unsigned long long c = 0;
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * stride_words; i < words; i += (size_t)gridDim.x * blockDim.x * stride_words) c += p[i];
    if (c == 0x123456789abcull) atomicAdd(sink, c);
}
__global__ void check(const uint32_t* p, size_t n, uint32_t v, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != (v ^ (uint32_t)i);
    if (c) atomicAdd(bad, c);
}
struct Range { char* va = nullptr; size_t chunk = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static hipError_t map_all(Range& r, int dev, size_t n) {       // back [va, va + n * chunk) with fresh physical chunks
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h; hipError_t e = hipMemCreate(&h, r.chunk, &prop, 0); if (e != hipSuccess) return e;
        e = hipMemMap(r.va + i * r.chunk, r.chunk, 0, h, 0); if (e != hipSuccess) return e;
        r.h.push_back(h);
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(r.va, n * r.chunk, &acc, 1);
}
static void unmap_all(Range& r) { for (size_t i = 0; i < r.h.size(); ++i) { (void)hipMemUnmap(r.va + i * r.chunk, r.chunk); (void)hipMemRelease(r.h[i]); } r.h.clear(); }

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 200;
    const size_t chunk = (size_t)(argc > 3 ? atoi(argv[3]) : 2) << 20, n = argc > 4 ? atoi(argv[4]) : 6, bytes = chunk * n, words = bytes / 4;
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* d_bad; CK(hipMalloc(&d_bad, 8));
    std::vector<uint32_t> host(words);
    uint32_t* big = nullptr; const size_t big_words = (size_t)16 << 28;      // mode 4: 16 GiB
    if (mode == 4) { CK(hipMalloc(&big, big_words * 4)); CK(hipMemset(big, 0, big_words * 4)); }
    Range many; many.chunk = (size_t)2 << 20; const size_t many_n = 6144;      // mode 7: 12 GiB in 6144 chunks of their own
    if (mode == 7) { void* b7 = nullptr; CK(hipMemAddressReserve(&b7, many_n * many.chunk, 0, nullptr, 0)); many.va = (char*)b7; CK(map_all(many, 0, many_n)); CK(hipMemset(many.va, 0, many_n * many.chunk)); }
    int same_va = 0, bad_kernel = 0, bad_d2h = 0, bad_after_h2d = 0, first_bad = -1, third = 0, third_kernel_ok = 0, third_copy_ok = 0;
    if (mode == 3) {
        Range cur; cur.chunk = chunk; size_t cur_n = n;
        void* base = nullptr; CK(hipMemAddressReserve(&base, cur_n * chunk, 0, nullptr, 0)); cur.va = (char*)base; CK(map_all(cur, 0, cur_n));
        std::vector<char*> seen;                                  // addresses of ranges that were freed
        for (int it = 0; it < iters; ++it) {
            const uint32_t b = 0x5B000000u + it;
            void* junk = nullptr; CK(hipMalloc(&junk, (size_t)(1 + it % 7) << 20));
            const size_t wn = n + 1 + (size_t)(it * 7 % 5);         // the wide range: n + 1 ... n + 5 chunks
            Range wide; wide.chunk = chunk;
            CK(hipMemAddressReserve(&base, wn * chunk, 0, nullptr, 0)); wide.va = (char*)base;
            for (char* q : seen) if (q == wide.va) { ++same_va; break; }
            CK(map_all(wide, 0, wn));
            const size_t w = wn * chunk / 4;
            fill<<<1024, 256, 0, s>>>((uint32_t*)wide.va, w, b);
            CK(hipMemsetAsync(d_bad, 0, 8, s));
            check<<<1024, 256, 0, s>>>((const uint32_t*)wide.va, w, b, d_bad);
            unsigned long long bad = 0; CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            host.resize(w); CK(hipMemcpy(host.data(), wide.va, w * 4, hipMemcpyDeviceToHost));
            size_t bad_h = 0; for (size_t i = 0; i < w; ++i) bad_h += host[i] != (b ^ (uint32_t)i);
            if ((bad || bad_h) && first_bad < 0) { first_bad = it; fprintf(stderr, "iter %d: wide range %p (%zu chunks): kernel-read bad words %llu, D2H bad words %zu, first word 0x%08x expected 0x%08x\n", it, (void*)wide.va, wn, bad, bad_h, host[0], b); }
            bad_kernel += bad != 0; bad_d2h += bad_h != 0;
            for (size_t i = 1; i < wn; i += 2) { (void)hipMemUnmap(wide.va + i * chunk, chunk); (void)hipMemRelease(wide.h[i]); }      // trim
            unmap_all(cur); CK(hipMemAddressFree(cur.va, cur_n * chunk)); seen.push_back(cur.va);                                  // retire the old one
            cur = wide; cur.h.clear(); cur_n = wn;
            for (size_t i = 0; i < wn; i += 2) { (void)hipMemUnmap(wide.va + i * chunk, chunk); (void)hipMemRelease(wide.h[i]); }
            CK(map_all(cur, 0, cur_n));                               // full again: the next round's "current range"
            CK(hipFree(junk));
        }
        printf("{\"mode\": 3, \"iterations\": %d, \"chunk_MiB\": %zu, \"chunks\": %zu, \"reservation_on_a_freed_address\": %d, \"bad_kernel_read\": %d, \"bad_memcpy_d2h\": %d, \"first_bad_iteration\": %d}\n",
               iters, chunk >> 20, n, same_va, bad_kernel, bad_d2h, first_bad);
        return 0;
    }
    for (int it = 0; it < iters; ++it) {
        const uint32_t a = 0xA5000000u + it, b = 0x5B000000u + it, c = 0xC3000000u + it;
        void* junk = nullptr; if (mode != 6) CK(hipMalloc(&junk, (size_t)(1 + it % 7) << 20));     // allocation churn, as in the test
        Range r; r.chunk = chunk;
        void* base = nullptr; CK(hipMemAddressReserve(&base, bytes, 0, nullptr, 0)); r.va = (char*)base;
        CK(map_all(r, 0, n));
        fill<<<1024, 256, 0, s>>>((uint32_t*)r.va, words, a);
        CK(hipStreamSynchronize(s));
        if (mode == 2) CK(hipDeviceSynchronize());
        unmap_all(r);
        char* old_va = r.va;
        if (mode == 4) { thrash<<<256, 256, 0, s>>>(big, big_words, 16384, d_bad); CK(hipStreamSynchronize(s)); }
        if (mode == 7) { for (int r3 = 0; r3 < 3; ++r3) thrash<<<256, 256, 0, s>>>((const uint32_t*)many.va, many_n * many.chunk / 4, many.chunk / 4, d_bad); CK(hipStreamSynchronize(s)); }
        if (mode == 5) { timespec ts = {0, 200000000}; nanosleep(&ts, nullptr); }
        if (mode != 1 && mode != 4 && mode != 5 && mode != 6 && mode != 7) { CK(hipMemAddressFree(r.va, bytes)); CK(hipMemAddressReserve(&base, bytes, 0, old_va, 0)); r.va = (char*)base; }
        same_va += r.va == old_va;
        if (junk) CK(hipFree(junk));
        CK(map_all(r, 0, n));
        fill<<<1024, 256, 0, s>>>((uint32_t*)r.va, words, b);
        CK(hipMemsetAsync(d_bad, 0, 8, s));
        check<<<1024, 256, 0, s>>>((const uint32_t*)r.va, words, b, d_bad);
        unsigned long long bad = 0; CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(host.data(), r.va, bytes, hipMemcpyDeviceToHost));
        size_t bad_h = 0; for (size_t i = 0; i < words; ++i) bad_h += host[i] != (b ^ (uint32_t)i);
        const uint32_t seen0 = host[0];
        for (size_t i = 0; i < words; ++i) host[i] = c ^ (uint32_t)i;
        CK(hipMemcpy(r.va, host.data(), bytes, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(d_bad, 0, 8, s));
        check<<<1024, 256, 0, s>>>((const uint32_t*)r.va, words, c, d_bad);
        unsigned long long bad2 = 0; CK(hipMemcpyAsync(&bad2, d_bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        if ((bad || bad_h || bad2) && first_bad < 0) { first_bad = it; fprintf(stderr, "iter %d: kernel-read bad words %llu, D2H bad words %zu (first word 0x%08x, expected 0x%08x), after H2D %llu\n", it, bad, bad_h, seen0, b, bad2); }
        bad_kernel += bad != 0; bad_d2h += bad_h != 0; bad_after_h2d += bad2 != 0;
        if (bad || bad_h || bad2) {     // a third view: chunk 0's physical memory mapped at a fresh address nobody has used.  What does it hold --
            // b (the kernel's fill reached it: the copy engine's view of the reused address was the stale one) or c (the H2D
            // copy reached it: the kernel's translation of the reused address was stale)?
            (void)hipMemUnmap(r.va, chunk);
            void* fresh = nullptr; CK(hipMemAddressReserve(&fresh, chunk, 0, nullptr, 0));
            CK(hipMemMap(fresh, chunk, 0, r.h[0], 0));
            hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(fresh, chunk, &acc, 1));
            uint32_t w0 = 0; CK(hipMemcpy(&w0, fresh, 4, hipMemcpyDeviceToHost));
            const char* what = w0 == b ? "b: kernel writes arrived, the COPY path was stale" : w0 == c ? "c: copies arrived, the KERNEL's translation was stale" : "neither";
            if (third < 3) fprintf(stderr, "iter %d: physical chunk 0 seen through a fresh address holds 0x%08x (%s)\n", it, w0, what);
            third_kernel_ok += w0 == b; third_copy_ok += w0 == c; ++third;
            CK(hipMemUnmap(fresh, chunk)); CK(hipMemAddressFree(fresh, chunk));
            CK(hipMemMap(r.va, chunk, 0, r.h[0], 0));      // put it back for the common cleanup
        }
        if (mode == 2) CK(hipDeviceSynchronize());
        unmap_all(r);
        CK(hipMemAddressFree(r.va, bytes));
    }
    printf("{\"mode\": %d, \"iterations\": %d, \"chunk_MiB\": %zu, \"chunks\": %zu, \"same_va\": %d, \"bad_kernel_read\": %d, \"bad_memcpy_d2h\": %d, \"bad_kernel_read_after_h2d\": %d, \"first_bad_iteration\": %d, "
           "\"third_view\": {\"cases\": %d, \"kernel_writes_arrived\": %d, \"copies_arrived\": %d}}\n",
           mode, iters, chunk >> 20, n, same_va, bad_kernel, bad_d2h, bad_after_h2d, first_bad, third, third_kernel_ok, third_copy_ok);
    return 0;
}
