// vmm_fault_injection.cpp -- VERDICT r04 next-3: the host code of the virtual-memory layer (proteus_amd/csrc/dswx_vmm.h:
// VaPool, VmRange::create / rehome / destroy, the chunk pool, dswx_vmm::pool_trim) under -fsanitize=address,undefined on
// the CPU, linked against a FAKE of the HIP virtual-memory calls that lives in this file and can fail the k-th call.
//
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ \
//       -I/opt/rocm/include -I proteus_amd/csrc tests/native/vmm_fault_injection.cpp -o tests/native/_build/vmm_fault_injection
//   (tests/test_vmm_fault_injection.py builds and runs it; no GPU, no libamdhip64)
//
// The fake keeps the state a real driver keeps -- reservations, physical allocations, mappings, access flags -- and is
// STRICT: it records a violation for every call a correct client never makes (map outside a reservation or over a live
// mapping, unmap of something not mapped, release of a dead handle, free of a reservation that still has mappings,
// access to unmapped addresses) and for the one rule this stack adds (dswx_vmm.h "address space"): an address a kernel
// has used must never be mapped onto OTHER physical memory.  "Kernels" are the harness calling touch() on a range.
//
// The scenario is the life of a placed batch and its successor: create a range, use it, create the wide range, use it,
// rehome the chosen chunks, use them, drop both old ranges (pool), build a second batch's range from the pool, trim
// while it is live, drop it, trim again.  It runs once without faults to count the HIP calls (N), then for every k in
// 1..N with the k-th call failing once, and again with every call from the k-th on failing (the undo paths under a dead
// driver).  After every run: no violation, the library's account (live / retired / loose / pooled) equals what the fake
// sees, and at the end no physical allocation is left that the library does not account for.  ASan's leak check covers
// the library's own heap (VmRange objects, vectors) and the fake's allocation records.
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

std::mutex& dswx_va_mutex() { static std::mutex* m = new std::mutex; return *m; }

#include "dswx_vmm.h"

// ---------------------------------------------------------------------------------------------------------- fake HIP
namespace fake {

struct Alloc { size_t size; int maps; bool released; uint64_t serial; };
struct Mapping { size_t size; Alloc* a; bool access; };

struct State {
    std::map<uintptr_t, size_t> reservations;
    std::set<Alloc*> allocs;                            // every record not yet freed (released AND unmapped frees it)
    std::map<uintptr_t, Mapping> mappings;
    std::map<uintptr_t, uint64_t> touched;              // chunk-start address -> serial of the memory a kernel saw there
    std::vector<std::string> violations;
    uintptr_t next_va = 0x7f0000000000ull;
    uint64_t next_serial = 1;
    long calls = 0, fail_at = -1;
    bool sticky = false;
    long failed = 0;
    hipError_t last = hipSuccess;
};
State* g = nullptr;

void violation(const char* fmt, ...) {
    char buf[300];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g->violations.push_back(buf);
}

// the k-th call fails (once, or from then on)
bool inject() {
    ++g->calls;
    if (g->fail_at > 0 && (g->calls == g->fail_at || (g->sticky && g->calls > g->fail_at))) {
        ++g->failed;
        g->last = hipErrorOutOfMemory;
        return true;
    }
    return false;
}

const std::pair<const uintptr_t, size_t>* reservation_of(uintptr_t va, size_t size) {
    auto it = g->reservations.upper_bound(va);
    if (it == g->reservations.begin()) return nullptr;
    --it;
    return (va >= it->first && va + size <= it->first + it->second) ? &*it : nullptr;
}

bool overlaps_mapping(uintptr_t va, size_t size) {
    for (const auto& m : g->mappings)
        if (va < m.first + m.second.size && m.first < va + size) return true;
    return false;
}

void drop_if_dead(Alloc* a) {
    if (a->released && a->maps == 0) { g->allocs.erase(a); delete a; }
}

// a kernel reads and writes [va, va + size)
void touch(const void* p, size_t size) {
    const uintptr_t va = (uintptr_t)p;
    size_t covered = 0;
    for (auto& m : g->mappings) {
        if (!(va < m.first + m.second.size && m.first < va + size)) continue;
        if (!m.second.access) violation("kernel access to %#" PRIxPTR " without hipMemSetAccess", m.first);
        g->touched[m.first] = m.second.a->serial;
        const uintptr_t lo = va > m.first ? va : m.first, hi = (va + size < m.first + m.second.size) ? va + size : m.first + m.second.size;
        covered += hi - lo;
    }
    if (covered != size) violation("kernel access to unmapped addresses in [%#" PRIxPTR ", +%zu)", va, size);
}

uint64_t physical_bytes() {
    uint64_t n = 0;
    for (const Alloc* a : g->allocs) n += a->size;
    return n;
}
uint64_t reserved_bytes() {
    uint64_t n = 0;
    for (const auto& r : g->reservations) n += r.second;
    return n;
}

}  // namespace fake

extern "C" {

hipError_t hipGetLastError(void) { const hipError_t e = fake::g->last; fake::g->last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "out of memory (injected)"; }

hipError_t hipMemAddressReserve(void** ptr, size_t size, size_t, void* addr, unsigned long long) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    if (size == 0) { violation("hipMemAddressReserve of 0 bytes"); return hipErrorInvalidValue; }
    uintptr_t va = (uintptr_t)addr;
    bool free_there = va != 0;
    if (va)
        for (const auto& r : g->reservations) free_there = free_there && !(va < r.first + r.second && r.first < va + size);
    if (!free_there) { va = g->next_va; g->next_va += (size + 0xfffffull) & ~0xfffffull; }
    g->reservations[va] = size;
    *ptr = (void*)va;
    return hipSuccess;
}

hipError_t hipMemAddressFree(void* devPtr, size_t size) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    auto it = g->reservations.find((uintptr_t)devPtr);
    if (it == g->reservations.end() || it->second != size) {
        violation("hipMemAddressFree(%p, %zu): no such reservation", devPtr, size);
        return hipErrorInvalidValue;
    }
    if (overlaps_mapping(it->first, it->second)) violation("hipMemAddressFree(%p): the reservation still has mappings", devPtr);
    g->reservations.erase(it);
    return hipSuccess;
}

hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* handle, size_t size, const hipMemAllocationProp*, unsigned long long) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    Alloc* a = new Alloc{size, 0, false, g->next_serial++};
    g->allocs.insert(a);
    *handle = (hipMemGenericAllocationHandle_t)a;
    return hipSuccess;
}

hipError_t hipMemRelease(hipMemGenericAllocationHandle_t handle) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    Alloc* a = (Alloc*)handle;
    if (!g->allocs.count(a) || a->released) { violation("hipMemRelease of a dead handle %p", (void*)handle); return hipErrorInvalidValue; }
    a->released = true;
    drop_if_dead(a);
    return hipSuccess;
}

hipError_t hipMemMap(void* ptr, size_t size, size_t offset, hipMemGenericAllocationHandle_t handle, unsigned long long) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    Alloc* a = (Alloc*)handle;
    const uintptr_t va = (uintptr_t)ptr;
    if (!g->allocs.count(a) || a->released) { violation("hipMemMap of a dead handle %p", (void*)handle); return hipErrorInvalidValue; }
    if (offset != 0 || size != a->size) { violation("hipMemMap: size / offset do not match the allocation"); return hipErrorInvalidValue; }
    if (!reservation_of(va, size)) { violation("hipMemMap(%p, %zu): outside every reservation", ptr, size); return hipErrorInvalidValue; }
    if (overlaps_mapping(va, size)) { violation("hipMemMap(%p): addresses already mapped", ptr); return hipErrorInvalidValue; }
    auto t = g->touched.find(va);
    if (t != g->touched.end() && t->second != a->serial)
        violation("hipMemMap(%p): an address a kernel has used is mapped onto OTHER physical memory (stale translations)", ptr);
    g->mappings[va] = Mapping{size, a, false};
    ++a->maps;
    return hipSuccess;
}

hipError_t hipMemUnmap(void* ptr, size_t size) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    auto it = g->mappings.find((uintptr_t)ptr);
    if (it == g->mappings.end() || it->second.size != size) { violation("hipMemUnmap(%p, %zu): not a mapping", ptr, size); return hipErrorInvalidValue; }
    Alloc* a = it->second.a;
    g->mappings.erase(it);
    --a->maps;
    drop_if_dead(a);
    return hipSuccess;
}

hipError_t hipMemSetAccess(void* ptr, size_t size, const hipMemAccessDesc*, size_t) {
    using namespace fake;
    if (inject()) return hipErrorOutOfMemory;
    const uintptr_t va = (uintptr_t)ptr;
    size_t covered = 0;
    for (auto& m : g->mappings)
        if (m.first >= va && m.first + m.second.size <= va + size) { m.second.access = true; covered += m.second.size; }
    if (covered != size) { violation("hipMemSetAccess(%p, %zu): range not fully mapped", ptr, size); return hipErrorInvalidValue; }
    return hipSuccess;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------- the harness
using dswx_vmm::VaPool;
using dswx_vmm::VmRange;
using dswx_vmm::va_pool;

namespace {

constexpr size_t kChunk = 2u << 20;
int g_failures = 0;
std::string g_tag;

#define CHECK(cond, ...)                                                          \
    do {                                                                          \
        if (!(cond)) {                                                            \
            ++g_failures;                                                         \
            fprintf(stderr, "FAIL [%s] %s:%d: ", g_tag.c_str(), __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                                         \
            fprintf(stderr, "\n");                                                \
        }                                                                         \
    } while (0)

void reset_pool() {
    VaPool& p = va_pool();
    p = VaPool();
}

// the library's account against what the fake driver sees
void check_account(const std::vector<const VmRange*>& live, const char* where) {
    VaPool& p = va_pool();
    uint64_t live_reserved = 0, live_mapped = 0;
    for (const VmRange* r : live)
        if (r) { live_reserved += r->reserved; live_mapped += r->mapped_bytes(); }
    CHECK(p.live == live_reserved, "%s: account live %" PRIu64 " != ranges in use %" PRIu64, where, p.live, live_reserved);
    uint64_t pooled = 0;
    for (const auto& s : p.spare) pooled += s.chunk;
    CHECK(p.pooled == pooled, "%s: account pooled %" PRIu64 " != spare chunks %" PRIu64, where, p.pooled, pooled);
    // every reservation the driver holds is a live range, a retired range, or (never) lost
    CHECK(fake::reserved_bytes() == p.live + p.retired, "%s: driver reservations %" PRIu64 " != live %" PRIu64 " + retired %" PRIu64,
          where, fake::reserved_bytes(), p.live, p.retired);
    // every physical allocation is mapped in a live range, in the pool, or accounted as lost to a failed call
    CHECK(fake::physical_bytes() == live_mapped + p.pooled + p.leaked, "%s: driver memory %" PRIu64 " != mapped in live ranges %" PRIu64
          " + pooled %" PRIu64 " + leaked %" PRIu64, where, fake::physical_bytes(), live_mapped, p.pooled, p.leaked);
    for (const std::string& v : fake::g->violations) CHECK(false, "%s: driver-rule violation: %s", where, v.c_str());
    fake::g->violations.clear();
}

void use(const VmRange* r) {                 // a kernel over every mapped chunk of the range
    for (size_t i = 0; i < r->handle.size(); ++i)
        if (r->mapped[i]) fake::touch(r->va + i * r->chunk, r->chunk);
}

void drop(VmRange*& r) {
    if (r) { r->destroy(); delete r; r = nullptr; }
}

// The life of a placed batch and its successor.  Returns the number of HIP calls made.
long scenario(long fail_at, bool sticky) {
    fake::State st;
    st.fail_at = fail_at;
    st.sticky = sticky;
    fake::g = &st;
    reset_pool();
    char tag[64];
    snprintf(tag, sizeof tag, "k=%ld%s", fail_at, sticky ? " sticky" : "");
    g_tag = tag;

    VmRange* first = new VmRange();          // dswx_batch_create(SLIDING): the output region, 3 chunks
    if (first->create(0, 3 * kChunk - 100, kChunk) != hipSuccess) {
        CHECK(!first->why.empty(), "create failed without a reason");
        CHECK(first->va == nullptr && first->reserved == 0 && first->handle.empty(), "a failed create left the range populated");
        delete first;
        first = nullptr;
    }
    check_account({first}, "after create");
    if (first) use(first);

    VmRange* wide = nullptr;                 // dswx_batch_place_slide: the wide range, 6 chunks, beside the first
    if (first) {
        wide = new VmRange();
        if (wide->create(0, 6 * kChunk, kChunk) != hipSuccess) { delete wide; wide = nullptr; }
        check_account({first, wide}, "after the wide create");
    }
    VmRange* home = nullptr;
    if (wide) {
        use(wide);
        // keep chunks 1, 2 and 4 (two intervals; chunk 3 stays behind as a hole)
        size_t base = 0;
        home = wide->rehome({{1 * kChunk + 5, 3 * kChunk - 7}, {4 * kChunk, 4 * kChunk + 1}}, &base);
        if (home) {
            CHECK(base == 1 * kChunk, "rehome base %zu", base);
            CHECK(home->reserved == 4 * kChunk && home->mapped_bytes() == 3 * kChunk, "home: reserved %zu mapped %zu", home->reserved,
                  home->mapped_bytes());
            CHECK(wide->mapped_bytes() == 3 * kChunk, "wide keeps %zu bytes after the move", wide->mapped_bytes());
        } else {
            CHECK(!wide->why.empty(), "rehome failed without a reason");
        }
        check_account({first, wide, home}, "after rehome");
        if (home) use(home);
        else if (wide->mapped_bytes() == 6 * kChunk) use(wide);      // the undo put every chunk back: the planes are usable
    }
    if (home) {                              // the placement is kept: the first-come range and the wide one are dropped
        drop(first);
        drop(wide);
        check_account({home}, "after dropping the first-come and the wide range");
        CHECK(va_pool().pooled + va_pool().leaked >= 6 * kChunk || fail_at > 0, "pool holds %" PRIu64, va_pool().pooled);
    } else {
        drop(wide);
        check_account({first}, "after dropping the wide range");
    }
    // a second batch arrives while the first is live: built from the pool before any new memory
    const uint64_t physical_before = fake::physical_bytes();
    VmRange* second = new VmRange();
    if (second->create(0, 2 * kChunk, kChunk) != hipSuccess) { delete second; second = nullptr; }
    if (second && fail_at < 0) CHECK(fake::physical_bytes() == physical_before, "the second range did not come from the pool");
    check_account({first, home, second}, "after the second create");
    if (second) use(second);

    // dswx_batch_pool_trim while ranges are live
    (void)dswx_vmm::pool_trim();
    check_account({first, home, second}, "after the trim with live ranges");
    if (home) use(home);
    if (first) use(first);
    if (second) use(second);

    drop(second);
    drop(home);
    drop(first);
    check_account({}, "after dropping everything");
    (void)dswx_vmm::pool_trim();
    check_account({}, "after the last trim");
    if (fail_at < 0) {
        CHECK(fake::physical_bytes() == 0, "memory left after the last trim: %" PRIu64, fake::physical_bytes());
        CHECK(va_pool().retired == fake::reserved_bytes() && va_pool().loose == 0, "retired %" PRIu64 " vs reserved %" PRIu64,
              va_pool().retired, fake::reserved_bytes());
    } else if (!sticky) {
        // one failed call may cost what the account says it cost, nothing more
        CHECK(fake::physical_bytes() == va_pool().leaked, "memory left %" PRIu64 " != leaked %" PRIu64, fake::physical_bytes(), va_pool().leaked);
    }
    // the fake's teardown (a process exit): whatever the library deliberately keeps reserved / lost goes with the process
    for (fake::Alloc* a : st.allocs) delete a;
    st.allocs.clear();
    const long calls = st.calls;
    fake::g = nullptr;
    return calls;
}

// the address-space budget (dswx_batch_va_budget): live + retired + a new request may not pass it, a refused request
// reserves nothing and says why, and chunk sizes are the ten powers of two
void budget_and_chunk_sizes() {
    fake::State st;
    fake::g = &st;
    reset_pool();
    g_tag = "budget";
    va_pool().budget = 5 * kChunk;
    VmRange a, b, c;
    CHECK(a.create(0, 3 * kChunk, kChunk) == hipSuccess, "first range refused: %s", a.why.c_str());
    const long calls = st.calls;
    CHECK(b.create(0, 3 * kChunk, kChunk) != hipSuccess && b.why.find("address-space budget") != std::string::npos, "why: %s", b.why.c_str());
    CHECK(st.calls == calls && b.va == nullptr, "a refused request reached the driver");
    a.destroy();                                    // retired: the addresses still count
    CHECK(c.create(0, 3 * kChunk, kChunk) != hipSuccess, "retired addresses were not counted against the budget");
    CHECK(c.create(0, 2 * kChunk, kChunk) == hipSuccess && fake::physical_bytes() == 3 * kChunk, "the pool was not used: %" PRIu64,
          fake::physical_bytes());
    check_account({&c}, "budget");
    uint64_t budget = 0, live = 0, retired = 0, loose = 0, pooled = 0;
    dswx_vmm::account(0, &budget, &live, &retired, &loose, &pooled);
    CHECK(budget == 5 * kChunk && live == 2 * kChunk && retired == 3 * kChunk && loose == 0 && pooled == kChunk, "account()");
    c.destroy();
    (void)dswx_vmm::pool_trim();
    check_account({}, "budget, after the trim");
    CHECK(fake::physical_bytes() == 0, "memory left");
    std::set<size_t> sizes;
    for (size_t bytes = 1; bytes < (size_t(64) << 30); bytes += bytes / 3 + 1) {
        const size_t c2 = dswx_vmm::chunk_for(bytes);
        sizes.insert(c2);
        CHECK((c2 & (c2 - 1)) == 0 && c2 >= (size_t(2) << 20) && c2 <= (size_t(1) << 30), "chunk_for(%zu) = %zu", bytes, c2);
        CHECK(c2 == (size_t(2) << 20) || c2 == (size_t(1) << 30) || (c2 >= bytes / 8 && c2 / 2 < bytes / 8), "chunk_for(%zu) = %zu", bytes, c2);
    }
    CHECK(sizes.size() == 10, "%zu chunk sizes", sizes.size());
    fake::g = nullptr;
}

}  // namespace

int main() {
    budget_and_chunk_sizes();
    const long n = scenario(-1, false);
    if (g_failures) { fprintf(stderr, "the fault-free run already fails\n"); return 1; }
    long injected = 0;
    for (int sticky = 0; sticky < 2; ++sticky)
        for (long k = 1; k <= n; ++k) {
            scenario(k, sticky != 0);
            ++injected;
        }
    printf("{\"hip_calls_fault_free\": %ld, \"runs_with_a_fault\": %ld, \"failures\": %d}\n", n, injected, g_failures);
    return g_failures ? 1 : 0;
}
