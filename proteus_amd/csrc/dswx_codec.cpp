// dswx_codec.cpp -- libdswx_codec.so: DEFLATE of GeoTIFF blocks on a pool of host threads (include/dswx_codec.h).
// Host-only C++17; links libz, loads libdeflate at run time when the system has it (no header needed: six
// prototypes of its stable public API are declared here).
#include "dswx_codec.h"

#include <dlfcn.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

thread_local char g_error[256] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

// ---- libdeflate, if present ---------------------------------------------------------------------------------
struct LibDeflate {
    void* (*alloc_compressor)(int) = nullptr;
    size_t (*zlib_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    void (*free_compressor)(void*) = nullptr;
    void* (*alloc_decompressor)() = nullptr;
    int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_decompressor)(void*) = nullptr;
    bool ok = false;
};

std::once_flag g_engine_once;
LibDeflate g_ld;
std::atomic<int> g_force_zlib{0};

void load_engine() {
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    LibDeflate l;
    l.alloc_compressor = reinterpret_cast<void* (*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
    l.zlib_compress = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(h, "libdeflate_zlib_compress"));
    l.free_compressor = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_compressor"));
    l.alloc_decompressor = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
    l.zlib_decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_zlib_decompress"));
    l.free_decompressor = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
    l.ok = l.alloc_compressor && l.zlib_compress && l.free_compressor && l.alloc_decompressor && l.zlib_decompress &&
           l.free_decompressor;
    if (l.ok) g_ld = l;
}

bool use_libdeflate() {
    std::call_once(g_engine_once, load_engine);
    return g_ld.ok && !g_force_zlib.load(std::memory_order_relaxed);
}

// per-thread codec objects (libdeflate's are not thread-safe; allocation is the expensive part)
struct ThreadCodecs {
    void* comp = nullptr;
    int comp_level = -1;
    void* decomp = nullptr;
    ~ThreadCodecs() {
        if (comp) g_ld.free_compressor(comp);
        if (decomp) g_ld.free_decompressor(decomp);
    }
};
thread_local ThreadCodecs t_codecs;

int deflate_one(const void* src, size_t n, void* dst, size_t cap, size_t* out, int level) {
    if (use_libdeflate()) {
        ThreadCodecs& c = t_codecs;
        if (!c.comp || c.comp_level != level) {
            if (c.comp) g_ld.free_compressor(c.comp);
            c.comp = g_ld.alloc_compressor(level);
            c.comp_level = level;
            if (!c.comp) return fail(DSWX_CODEC_ERR_ARG, "libdeflate_alloc_compressor(%d) failed", level);
        }
        const size_t got = g_ld.zlib_compress(c.comp, src, n, dst, cap);
        if (got == 0) return fail(DSWX_CODEC_ERR_SPACE, "compressed block does not fit %zu bytes", cap);
        *out = got;
        return DSWX_CODEC_OK;
    }
    uLongf len = static_cast<uLongf>(cap);
    const int rc = compress2(static_cast<Bytef*>(dst), &len, static_cast<const Bytef*>(src), static_cast<uLong>(n), level);
    if (rc == Z_BUF_ERROR) return fail(DSWX_CODEC_ERR_SPACE, "compressed block does not fit %zu bytes", cap);
    if (rc != Z_OK) return fail(DSWX_CODEC_ERR_DATA, "zlib compress2 failed (%d)", rc);
    *out = len;
    return DSWX_CODEC_OK;
}

int inflate_one(const void* src, size_t n, void* dst, size_t cap, size_t* out) {
    if (use_libdeflate()) {
        ThreadCodecs& c = t_codecs;
        if (!c.decomp) {
            c.decomp = g_ld.alloc_decompressor();
            if (!c.decomp) return fail(DSWX_CODEC_ERR_ARG, "libdeflate_alloc_decompressor failed");
        }
        size_t got = 0;
        const int rc = g_ld.zlib_decompress(c.decomp, src, n, dst, cap, &got);
        if (rc == 0) { *out = got; return DSWX_CODEC_OK; }
        if (rc == 3) return fail(DSWX_CODEC_ERR_SPACE, "block inflates to more than %zu bytes", cap);
        return fail(DSWX_CODEC_ERR_DATA, "corrupt zlib stream (libdeflate result %d)", rc);
    }
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit(&z) != Z_OK) return fail(DSWX_CODEC_ERR_DATA, "inflateInit failed");
    z.next_in = const_cast<Bytef*>(static_cast<const Bytef*>(src));
    z.avail_in = static_cast<uInt>(n);
    z.next_out = static_cast<Bytef*>(dst);
    z.avail_out = static_cast<uInt>(cap);
    const int rc = inflate(&z, Z_FINISH);
    const size_t got = z.total_out;
    inflateEnd(&z);
    if (rc == Z_STREAM_END) { *out = got; return DSWX_CODEC_OK; }
    if (rc == Z_BUF_ERROR || rc == Z_OK) return fail(DSWX_CODEC_ERR_SPACE, "block inflates to more than %zu bytes", cap);
    return fail(DSWX_CODEC_ERR_DATA, "corrupt zlib stream (zlib result %d)", rc);
}

// TIFF compression 5 (LZW, TIFF 6.0 section 13): codes of 9 .. 12 bits packed most significant bit first, 256 = clear the
// table, 257 = end of information, the code width grows one code EARLY (when entry 2^width - 1 would be assigned next) --
// what libtiff, hence GDAL's COMPRESS=LZW, writes.  A stream that starts like the pre-6.0 least-significant-bit-first variant
// (libtiff's "old-style LZW") is refused.  Like libtiff's decoder this stops when the block is full.
int unlzw_one(const void* src_v, size_t n, void* dst_v, size_t cap, size_t* out) {
    const unsigned char* src = static_cast<const unsigned char*>(src_v);
    unsigned char* dst = static_cast<unsigned char*>(dst_v);
    if (n >= 2 && src[0] == 0 && (src[1] & 1)) return fail(DSWX_CODEC_ERR_DATA, "old-style (LSB-first) LZW is not supported");
    // every table entry is a string that already stands in the output (entry k = the string of the previous code + the first
    // byte of the current one, which were written back to back): where it starts and how long it is
    size_t where[4096];
    uint16_t length[4096];
    int nbits = 9, next = 258, old = -1;
    uint64_t acc = 0;
    int have = 0;
    size_t in = 0, pos = 0, pos_old = 0;
    int len_old = 0;
    while (pos < cap) {
        while (have < nbits && in < n) { acc = (acc << 8) | src[in++]; have += 8; }
        if (have < nbits) break;                                   // ran out of codes without an end code: what there is, is the block
        const int code = (int)((acc >> (have - nbits)) & ((1u << nbits) - 1));
        have -= nbits;
        if (code == 257) break;
        if (code == 256) { nbits = 9; next = 258; old = -1; continue; }
        if (old < 0) {
            if (code > 255) return fail(DSWX_CODEC_ERR_DATA, "corrupt LZW stream (code %d after a clear code)", code);
            pos_old = pos;
            len_old = 1;
            dst[pos++] = (unsigned char)code;
            old = code;
            continue;
        }
        const bool added = next < 4096;
        if (added) { where[next] = pos_old; length[next] = (uint16_t)(len_old + 1); }
        if (code > (added ? next : next - 1))
            return fail(DSWX_CODEC_ERR_DATA, "corrupt LZW stream (code %d with %d entries)", code, next);
        const int len = code < 256 ? 1 : length[code];
        const size_t room = cap - pos, take = (size_t)len < room ? (size_t)len : room;
        if (code < 256) {
            dst[pos] = (unsigned char)code;
        } else {
            const size_t from = where[code];
            if (from + take <= pos) memcpy(dst + pos, dst + from, take);
            else for (size_t i = 0; i < take; ++i) dst[pos + i] = dst[from + i];     // the entry made just now overlaps its own copy
        }
        pos_old = pos;
        len_old = len;
        pos += take;
        if (added) {
            ++next;
            if (next > (1 << nbits) - 2 && nbits < 12) ++nbits;
        }
        old = code;
    }
    *out = pos;
    return DSWX_CODEC_OK;
}

// The processors this process may really use: the hardware threads, cut down to the container's CPU bandwidth quota
// (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  More runnable threads than that do not run faster: the quota is spent
// earlier in every period and the whole group is throttled until the next (measured on the MI355X box of this
// project: 256 hardware threads, quota 16 -> 1,740 blocks/s with 32 threads, 830 with 256).
std::atomic<int> g_budget_override{0};

int detected_cpu_budget() {
    static const int budget = [] {
        int n = (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        long long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = "";
            if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 100000; fclose(g); }
        }
        if (quota > 0 && period > 0) {
            const int cores = (int)((quota + period - 1) / period);
            if (cores >= 1 && cores < n) n = cores;
        }
        return n;
    }();
    return budget;
}

// the detected budget, or the share of it the caller was given (dswx_codec_set_cpu_budget: the node-level driver divides
// the container's processors among its worker processes)
int cpu_budget() {
    const int o = g_budget_override.load(std::memory_order_relaxed);
    const int d = detected_cpu_budget();
    return (o >= 1 && o < d) ? o : d;
}

// ---- the pool ------------------------------------------------------------------------------------------------
// One process-wide set of workers; a call is a Batch of n independent blocks that the workers and the calling thread
// take one at a time (atomic counter).  Several calls may be in flight at once (the product writes its layers side by
// side): a worker serves the oldest batch that still has blocks to hand out and fewer helpers than it asked for.
// A Batch is shared (shared_ptr): a worker may look at its counters after the owner has seen the last block done.
struct Batch {
    int n = 0;
    int want = 0;                       // helpers this call asked for (besides the caller)
    int inside = 0;                     // helpers working on it now (guarded by the pool mutex)
    std::atomic<int> next{0};
    std::atomic<int> done{0};
    std::atomic<int> status{0};
    char error[256] = "";
    std::mutex err_mutex;
    std::function<int(int)> work;       // only called for i < n, i.e. before done reaches n: may refer to the owner's stack
    std::mutex m;
    std::condition_variable cv;
};

void drain(Batch& b) {
    for (;;) {
        const int i = b.next.fetch_add(1);
        if (i >= b.n) return;
        const int rc = b.status.load() ? 0 : b.work(i);             // after a failure the rest is skipped
        if (rc) {
            std::lock_guard<std::mutex> lock(b.err_mutex);
            if (!b.status.load()) {
                snprintf(b.error, sizeof b.error, "%s", g_error);
                b.status.store(rc);
            }
        }
        if (b.done.fetch_add(1) + 1 >= b.n) {
            { std::lock_guard<std::mutex> g(b.m); }
            b.cv.notify_all();
        }
    }
}

class Pool {
public:
    static Pool& get() {
        static Pool* p = new Pool;      // never destroyed: detached workers outlive static destructors at exit
        return *p;
    }
    void run(const std::shared_ptr<Batch>& b, int threads) {
        const int helpers = threads > 1 ? (threads - 1 < b->n - 1 ? threads - 1 : b->n - 1) : 0;
        if (helpers > 0) {
            b->want = helpers;
            {
                std::lock_guard<std::mutex> lock(m_);
                // as many workers as the calls in flight ask for together, up to the machine's hardware threads: several
                // layers (and several tiles) are deflated side by side, each call bringing its own demand
                int demand = helpers;
                for (const auto& q : queue_) demand += q->want;
                const int cap = cpu_budget();       // (helpers + the calling threads: a little over the budget, never many times)
                while (n_workers_ < demand && n_workers_ < cap && n_workers_ < 1024) {
                    std::thread([this] { loop(); }).detach();
                    ++n_workers_;
                }
                queue_.push_back(b);
            }
            cv_.notify_all();
        }
        drain(*b);
        if (helpers > 0) {
            {
                std::lock_guard<std::mutex> lock(m_);
                for (auto it = queue_.begin(); it != queue_.end(); ++it)
                    if (it->get() == b.get()) { queue_.erase(it); break; }
            }
            std::unique_lock<std::mutex> lk(b->m);
            b->cv.wait(lk, [&] { return b->done.load() >= b->n; });
        }
    }

private:
    void loop() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            std::shared_ptr<Batch> b;
            for (const auto& q : queue_)
                if (q->next.load() < q->n && q->inside < q->want) { b = q; break; }
            if (!b) { cv_.wait(lk); continue; }
            ++b->inside;
            lk.unlock();
            drain(*b);
            lk.lock();
            --b->inside;
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Batch>> queue_;
    int n_workers_ = 0;
};

int run_blocks(int n, int threads, std::function<int(int)> work) {
    auto b = std::make_shared<Batch>();
    b->n = n;
    b->work = std::move(work);
    Pool::get().run(b, threads);
    if (b->status.load()) {
        snprintf(g_error, sizeof g_error, "%s", b->error);
        return b->status.load();
    }
    return DSWX_CODEC_OK;
}

}  // namespace

extern "C" {

int dswx_codec_abi_version(void) { return DSWX_CODEC_ABI_VERSION; }

const char* dswx_codec_engine(void) { return use_libdeflate() ? "libdeflate" : "zlib"; }

const char* dswx_codec_last_error(void) { return g_error; }

int dswx_codec_cpu_budget(void) { return cpu_budget(); }

int dswx_codec_set_cpu_budget(int processors) {
    g_budget_override.store(processors > 0 ? processors : 0);
    return DSWX_CODEC_OK;
}

int dswx_codec_force_zlib(int on) {
    g_force_zlib.store(on ? 1 : 0);
    return DSWX_CODEC_OK;
}

size_t dswx_codec_deflate_bound(size_t bytes) {
    // zlib's compressBound = n + n/4096 + n/16384 + n/33554432 + 13; libdeflate's bound is 5 bytes per 10 kB block + 1 + 9 + 6:
    // the larger of the two with room to spare
    return bytes + (bytes >> 9) + 64;
}

int dswx_codec_deflate_blocks(const void* const* src, const size_t* src_bytes, void* const* dst, const size_t* dst_cap,
                              size_t* dst_bytes, int32_t n, int32_t level, int32_t threads) {
    if (n < 0 || (n > 0 && (!src || !src_bytes || !dst || !dst_cap || !dst_bytes)))
        return fail(DSWX_CODEC_ERR_ARG, "NULL argument");
    if (level < 1 || level > 9) return fail(DSWX_CODEC_ERR_ARG, "level %d outside 1 .. 9", level);
    if (n == 0) return DSWX_CODEC_OK;
    return run_blocks(n, threads, [&](int i) {
        if (!src[i] && src_bytes[i]) return fail(DSWX_CODEC_ERR_ARG, "block %d is NULL", i);
        return deflate_one(src[i], src_bytes[i], dst[i], dst_cap[i], &dst_bytes[i], level);
    });
}

int dswx_codec_inflate_blocks(const void* const* src, const size_t* src_bytes, void* const* dst, const size_t* dst_cap,
                              size_t* dst_bytes, int32_t n, int32_t threads) {
    if (n < 0 || (n > 0 && (!src || !src_bytes || !dst || !dst_cap || !dst_bytes)))
        return fail(DSWX_CODEC_ERR_ARG, "NULL argument");
    if (n == 0) return DSWX_CODEC_OK;
    return run_blocks(n, threads, [&](int i) {
        if (!src[i] || !dst[i]) return fail(DSWX_CODEC_ERR_ARG, "block %d is NULL", i);
        return inflate_one(src[i], src_bytes[i], dst[i], dst_cap[i], &dst_bytes[i]);
    });
}

int dswx_codec_unlzw_blocks(const void* const* src, const size_t* src_bytes, void* const* dst, const size_t* dst_cap,
                            size_t* dst_bytes, int32_t n, int32_t threads) {
    if (n < 0 || (n > 0 && (!src || !src_bytes || !dst || !dst_cap || !dst_bytes)))
        return fail(DSWX_CODEC_ERR_ARG, "NULL argument");
    if (n == 0) return DSWX_CODEC_OK;
    return run_blocks(n, threads, [&](int i) {
        if (!src[i] || !dst[i]) return fail(DSWX_CODEC_ERR_ARG, "block %d is NULL", i);
        return unlzw_one(src[i], src_bytes[i], dst[i], dst_cap[i], &dst_bytes[i]);
    });
}

}  // extern "C"
