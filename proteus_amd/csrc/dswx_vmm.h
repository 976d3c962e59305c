// dswx_vmm.h -- the virtual-memory layer under the sliding placement (dswx_batch_place_slide): the process-wide account
// of address space, the pool of physical chunks, and VmRange, a reserved range backed chunk by chunk.  Header-only and
// HOST-only: it needs nothing but the HIP runtime API (<hip/hip_runtime_api.h>), so that tests/native/vmm_fault_injection.cpp
// can compile it with gcc under -fsanitize=address,undefined against an in-test fake of the hipMem* calls that fails the
// k-th call.  dswx_batch.hip is the only product user.
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

// One mutex for every device allocation of the library and for the account below (defined in dswx_hip.hip; the fault-
// injection test defines its own).
std::mutex& dswx_va_mutex();

namespace dswx_vmm {

// ---- address space ----------------------------------------------------------------------------------------------
// An address that a kernel has accessed through one mapping must never be mapped onto other physical memory in this
// process.  Measured in round 4 with plain HIP (tools/vmm_reuse_repro.hip, profiles/r04_vmm_reuse_repro.jsonl; ROCm 7.2,
// gfx950): map chunk A at VA, a kernel fills it, hipMemUnmap + hipMemRelease, map a NEW chunk B at the same VA -- with or
// without hipMemAddressFree / hipMemAddressReserve in between, with or without hipDeviceSynchronize before the unmap --
// and in 31 of 200 iterations the next kernel's stores never arrive in B (B, seen through a fresh address, still holds
// what a hipMemcpy put there) while hipMemcpy through VA reads and writes B: the KERNEL's translation of VA is stale
// (it still points at A's released memory); the copy path resolves VA afresh.  Addresses that were mapped but never
// touched by a kernel are safe to reuse (mode 3 of the reproducer: 0 of 200).  This is what round 3 saw as "layers read
// back zeroed" (8 - 27 of 80 two-placement cases) and fenced by retiring ranges; the same library built with the
// addresses freed (7 of 160 cases wrong), device-synchronised and freed (6 of 160) or kept in a free list and re-mapped
// (9 of 160) fails, retired it does not (0 of 160): tests/helpers/vmm_policy_trial.py, profiles/r04_vmm_policy_trial.json.
// So the addresses of a dropped range are never mapped again.  What happens to them and to the memory they held is shaped by
// a second property of this stack (tools/lab/vmm_meminfo.hip, profiles/r04_vmm_meminfo.json): the physical memory of a chunk
// that was ever mapped returns to the device only when the RESERVATION it was mapped in is freed (hipMemAddressFree) --
// hipMemUnmap + hipMemRelease alone keep it allocated.  Freeing the addresses and reserving them again at once, empty
// (a quarantine) returns the memory and gets the same addresses back (32 of 32 GiB in the probe) -- but between the two
// calls the addresses are up for grabs by any other thread of the process (a four-thread stress lost one range in a few
// hundred to an allocation the library cannot fence: the runtime's own, numpy's mmap, ...), and whoever maps GPU memory
// there inherits the stale translations.  Hence two steps:
//   * by default a dropped range is RETIRED: its chunks are unmapped and their physical memory goes into a process-wide
//     POOL (per device and chunk size) from which later ranges are built before any new memory is created; its addresses
//     stay reserved, empty.  Nothing is ever exposed; the memory stays with the library (dswx_batch_va_budget reports
//     pooled_bytes) and is reused by the next batch or placement of the same chunk size;
//   * dswx_batch_pool_trim() -- the caller's decision, for a moment when no other thread of the process allocates --
//     releases the pooled chunks and does the free + quarantine of every retired range: the memory goes back to the
//     device, the addresses stay out of circulation (a range whose addresses were lost in that instant is counted as
//     `loose`).
// Address space is consumed for good either way -- 100 - 160 GiB per placed batch at 256 tiles, of the 128 TiB a process
// has -- and the library keeps count (dswx_batch_va_budget, dswx_batch_info_t.va_*): beyond a budget it reserves no more,
// dswx_batch_create(DSWX_BATCH_SLIDING_OUTPUTS) falls back to the packed allocation and dswx_batch_place_slide leaves the
// planes where they are, both with the reason in dswx_batch_info_t.note.
// DSWX_VM_FREE_ADDRESSES (build-time, for tests/helpers/vmm_policy_trial.py only): 1 = release the chunks and hipMemAddressFree a
// dropped range, 2 = hipDeviceSynchronize first -- the two unsafe forms, kept so that the trial can be repeated on a newer ROCm.
#ifndef DSWX_VM_FREE_ADDRESSES
#define DSWX_VM_FREE_ADDRESSES 0
#endif

struct VaPool {                                         // guarded by dswx_va_mutex()
    uint64_t live = 0;                                  // reserved by ranges in use
    uint64_t retired = 0;                               // reserved by dropped ranges (empty, for good)
    uint64_t loose = 0;                                 // dropped ranges whose addresses were lost during a trim
    uint64_t leaked = 0;                                // physical memory the library can no longer return: a driver call
                                                        //   that cannot fail did (unmap / release); never re-used, never freed
    uint64_t budget = 64ull << 40;                      // live + retired may not pass this: half of the 47-bit space
    struct Spare { int device; size_t chunk; hipMemGenericAllocationHandle_t handle; };
    std::vector<Spare> spare;                           // physical chunks of dropped ranges, unmapped, for later ranges
    uint64_t pooled = 0;                                // their bytes
    std::vector<std::pair<char*, size_t>> untrimmed;    // retired ranges whose reservation still pins released memory
};
inline VaPool& va_pool() { static VaPool* p = new VaPool; return *p; }     // never destroyed: frees may arrive during exit

// A reserved range of the virtual address space backed chunk by chunk by physical allocations (HIP virtual memory
// management).  The sliding placement maps a range longer than the output planes, times the kernel with the planes at
// several places of it, and keeps only the chunks under the best one -- by moving those chunks (their physical memory,
// hipMemGenericAllocationHandle_t) into a fresh range and dropping the wide one.
// Life cycle: see "address space" above (retire + pool by default, dswx_batch_pool_trim for the memory).
struct VmRange {
    char* va = nullptr;
    size_t reserved = 0;       // the reservation = handle.size() * chunk
    size_t chunk = 0;
    int device = 0;
    std::string why;           // create() / rehome() failed: the reason, for dswx_batch_info_t.note
    bool damaged = false;      // a failed rehome() could not put every chunk back: drop the range, do not launch over it
    std::vector<hipMemGenericAllocationHandle_t> handle;
    std::vector<char> mapped;  // chunk i of the range is backed by handle[i]

    size_t mapped_bytes() const {
        size_t n = 0;
        for (char m : mapped) n += m ? chunk : 0;
        return n;
    }
    static hipMemAllocationProp prop_of(int dev) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        return prop;
    }
    hipError_t allow(size_t first_chunk, size_t n_chunks) {
        hipMemAccessDesc acc = {};
        acc.location = prop_of(device).location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        return hipMemSetAccess(va + first_chunk * chunk, n_chunks * chunk, &acc, 1);
    }
    void destroy() {
#if DSWX_VM_FREE_ADDRESSES == 2
        (void)hipDeviceSynchronize();
#endif
        std::lock_guard<std::mutex> lock(dswx_va_mutex());
        VaPool& pool = va_pool();
        bool stuck = false;                                             // a chunk could not be unmapped
        for (size_t i = 0; i < handle.size(); ++i)
            if (mapped[i]) {
                if (hipMemUnmap(va + i * chunk, chunk) != hipSuccess) {
                    // it stays mapped here for good: the range is then never freed (a reservation with a mapping in it
                    // must not be), the memory is written off
                    (void)hipGetLastError();
                    pool.leaked += chunk;
                    stuck = true;
                    continue;
                }
#if DSWX_VM_FREE_ADDRESSES
                (void)hipMemRelease(handle[i]);
#else
                pool.spare.push_back({device, chunk, handle[i]});       // the memory stays with the library: the next range's
                pool.pooled += chunk;
#endif
            }
        handle.clear();
        mapped.clear();
        if (va) {
            pool.live -= reserved;
#if DSWX_VM_FREE_ADDRESSES
            (void)hipMemAddressFree(va, reserved);
#else
            pool.retired += reserved;                                   // reserved, empty, for good
            if (!stuck) pool.untrimmed.push_back({va, reserved});
#endif
        }
        va = nullptr;
        reserved = 0;
    }
    // reserve `need` bytes of addresses (a multiple of the chunk size), nothing mapped
    hipError_t reserve(int dev, size_t need, size_t chunk_bytes) {
        device = dev;
        chunk = chunk_bytes;
        VaPool& pool = va_pool();
        std::lock_guard<std::mutex> lock(dswx_va_mutex());
        if (pool.live + pool.retired + need > pool.budget) {
            char buf[200];
            snprintf(buf, sizeof buf, "address-space budget: %llu bytes reserved by live ranges + %llu retired + %llu "
                     "wanted > %llu (dswx_batch_va_budget)", (unsigned long long)pool.live,
                     (unsigned long long)pool.retired, (unsigned long long)need, (unsigned long long)pool.budget);
            why = buf;
            return hipErrorOutOfMemory;
        }
        void* base = nullptr;
        const hipError_t e = hipMemAddressReserve(&base, need, 0, nullptr, 0);
        if (e != hipSuccess) {
            why = std::string("hipMemAddressReserve: ") + hipGetErrorString(e);
            return e;
        }
        va = static_cast<char*>(base);
        reserved = need;
        pool.live += reserved;
        handle.assign(need / chunk, hipMemGenericAllocationHandle_t{});
        mapped.assign(need / chunk, 0);
        return hipSuccess;
    }
    // reserve `bytes` (rounded up to whole chunks), back all of it, make it accessible from `dev`
    hipError_t create(int dev, size_t bytes, size_t chunk_bytes) {
        const size_t n = (bytes + chunk_bytes - 1) / chunk_bytes;
        hipError_t e = reserve(dev, n * chunk_bytes, chunk_bytes);
        if (e != hipSuccess) return e;
        const hipMemAllocationProp prop = prop_of(dev);
        for (size_t i = 0; i < n; ++i) {
            hipMemGenericAllocationHandle_t h;
            bool from_pool = false;
            {
                std::lock_guard<std::mutex> lock(dswx_va_mutex());
                VaPool& pool = va_pool();
                for (size_t k = pool.spare.size(); k-- > 0;)
                    if (pool.spare[k].device == dev && pool.spare[k].chunk == chunk) {
                        h = pool.spare[k].handle;
                        pool.spare.erase(pool.spare.begin() + (long)k);
                        pool.pooled -= chunk;
                        from_pool = true;
                        break;
                    }
            }
            e = from_pool ? hipSuccess : hipMemCreate(&h, chunk, &prop, 0);
            if (e == hipSuccess) {
                e = hipMemMap(va + i * chunk, chunk, 0, h, 0);
                if (e != hipSuccess && hipMemRelease(h) != hipSuccess) {
                    std::lock_guard<std::mutex> lock(dswx_va_mutex());
                    va_pool().leaked += chunk;
                }
            }
            if (e != hipSuccess) {
                why = std::string("hipMemCreate / hipMemMap: ") + hipGetErrorString(e);
                destroy();
                return e;
            }
            handle[i] = h;
            mapped[i] = 1;
        }
        e = allow(0, n);
        if (e != hipSuccess) {
            why = std::string("hipMemSetAccess: ") + hipGetErrorString(e);
            destroy();
        }
        return e;
    }
    // The chunks that touch one of the intervals [lo, hi) of this range, moved -- the same physical memory -- into a
    // fresh range that spans from the first to the last of them (holes stay unmapped); *base = offset of the new range's
    // first byte in this one.  This range keeps its other chunks: destroy() it afterwards.  nullptr: nothing was moved
    // (`why` says what failed).  The stream must be idle.
    VmRange* rehome(const std::vector<std::pair<size_t, size_t>>& keep, size_t* base) {
        const size_t n = handle.size();
        std::vector<char> used(n, 0);
        size_t first = n, last = 0;
        for (size_t i = 0; i < n; ++i) {
            const size_t c0 = i * chunk, c1 = c0 + chunk;
            for (const auto& iv : keep) used[i] = used[i] || (c0 < iv.second && iv.first < c1);
            if (used[i] && mapped[i]) { first = i < first ? i : first; last = i; }
        }
        if (first == n) { why = "rehome: nothing to keep"; return nullptr; }
        VmRange* home = new VmRange();
        if (home->reserve(device, (last - first + 1) * chunk, chunk) != hipSuccess) {
            why = home->why;
            delete home;
            return nullptr;
        }
        hipError_t e = hipSuccess;
        size_t moved = first;
        for (; moved <= last && e == hipSuccess; ++moved) {
            if (!(used[moved] && mapped[moved])) continue;
            e = hipMemUnmap(va + moved * chunk, chunk);
            if (e != hipSuccess) break;
            mapped[moved] = 0;
            e = hipMemMap(home->va + (moved - first) * chunk, chunk, 0, handle[moved], 0);
            if (e != hipSuccess) break;
            home->handle[moved - first] = handle[moved];
            home->mapped[moved - first] = 1;
            e = home->allow(moved - first, 1);
            if (e != hipSuccess) break;
        }
        if (e != hipSuccess) {
            // undo: every chunk back where it was (the addresses of THIS range have been used by kernels, but they get
            // their own physical memory back: the translations that may linger are the right ones)
            why = std::string("rehome: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            uint64_t lost = 0;
            bool home_stuck = false;
            for (size_t i = first; i <= last && i <= moved; ++i) {
                if (!used[i] || mapped[i]) continue;
                if (home->mapped[i - first]) {
                    if (hipMemUnmap(home->va + (i - first) * chunk, chunk) != hipSuccess) {
                        // still mapped in `home`: that range is then never freed, the chunk is written off
                        home_stuck = true;
                        damaged = true;
                        lost += chunk;
                        continue;
                    }
                    home->mapped[i - first] = 0;
                }
                if (hipMemMap(va + i * chunk, chunk, 0, handle[i], 0) == hipSuccess) {
                    mapped[i] = 1;                                  // back where it was (destroy() unmaps and pools it)
                    if (allow(i, 1) != hipSuccess) damaged = true;  // ... but not accessible
                } else {
                    damaged = true;                                 // gone: the planes over it are no longer valid
                    if (hipMemRelease(handle[i]) != hipSuccess) lost += chunk;
                }
            }
            (void)hipGetLastError();
            // `home` gave its chunks back (or is stuck with one it could not unmap): destroy() must neither pool nor unmap them
            {
                std::lock_guard<std::mutex> lock(dswx_va_mutex());
                VaPool& pool = va_pool();
                pool.leaked += lost;
                pool.live -= home->reserved;
                pool.retired += home->reserved;
                if (!home_stuck) pool.untrimmed.push_back({home->va, home->reserved});
            }
            home->va = nullptr;
            home->reserved = 0;
            home->handle.clear();
            home->mapped.clear();
            delete home;
            return nullptr;
        }
        *base = first * chunk;
        return home;
    }
};

// chunk size of a range that holds `bytes` of planes: about an eighth of the planes, as a POWER OF TWO between 2 MiB and
// 1 GiB -- ten sizes in all, so that the pooled chunks of one batch geometry serve the next one too (with
// sizes in 2 MiB steps nearly every geometry below 8 GiB had a pool of its own)
inline size_t chunk_for(size_t bytes) {
    size_t c = size_t(2) << 20;
    while (c < (size_t(1) << 30) && c < bytes / 8) c <<= 1;
    return c;
}

// (`leaked` -- memory written off after a driver call that cannot fail did -- is internal: the fault-injection test reads it)
inline void account(uint64_t new_budget_bytes, uint64_t* budget_bytes, uint64_t* live_bytes, uint64_t* retired_bytes,
                    uint64_t* loose_bytes, uint64_t* pooled_bytes) {
    VaPool& pool = va_pool();
    std::lock_guard<std::mutex> lock(dswx_va_mutex());
    if (new_budget_bytes) pool.budget = new_budget_bytes;
    if (budget_bytes) *budget_bytes = pool.budget;
    if (live_bytes) *live_bytes = pool.live;
    if (retired_bytes) *retired_bytes = pool.retired;
    if (loose_bytes) *loose_bytes = pool.loose;
    if (pooled_bytes) *pooled_bytes = pool.pooled;
}

// dswx_batch_pool_trim
inline uint64_t pool_trim() {
    VaPool& pool = va_pool();
    std::lock_guard<std::mutex> lock(dswx_va_mutex());
    uint64_t released = 0;
    for (const VaPool::Spare& sp : pool.spare) {            // (a handle carries its device: no current-device switch needed)
        if (hipMemRelease(sp.handle) == hipSuccess) released += sp.chunk;
        else { (void)hipGetLastError(); pool.leaked += sp.chunk; }
    }
    pool.spare.clear();
    pool.pooled = 0;
    // the reservations of the retired ranges still pin that memory: free each and take its addresses back at once, empty
    std::vector<std::pair<char*, size_t>> again_later;
    for (const auto& r : pool.untrimmed) {
        if (hipMemAddressFree(r.first, r.second) != hipSuccess) {      // still reserved: nothing is exposed, try again next time
            (void)hipGetLastError();
            again_later.push_back(r);
            continue;
        }
        void* again = nullptr;
        if (hipMemAddressReserve(&again, r.second, 0, r.first, 0) != hipSuccess || again != r.first) {
            if (again) (void)hipMemAddressFree(again, r.second);       // another thread took them in that instant
            (void)hipGetLastError();
            pool.retired -= r.second;
            pool.loose += r.second;
        }
    }
    pool.untrimmed.swap(again_later);
    return released;
}

}  // namespace dswx_vmm
