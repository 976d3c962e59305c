// codec_stress.cpp -- the thread pool of libdswx_codec.so (proteus_amd/csrc/dswx_codec.cpp, compiled INTO this harness) under
// ThreadSanitizer and under ASan + UBSan (tests/test_codec_sanitizers.py).  CPU only.
//
// Scenario: several caller threads (the product writes its layers side by side, several tiles in flight) issue
// deflate / inflate calls of random shapes at the same time -- batches of 1 ... 97 blocks of 1 ... 70,000 bytes, compressible,
// incompressible and empty blocks, 1 ... 12 threads per call, both engines (libdeflate when the system has it, zlib) --
// and check every round trip; calls that must fail (a destination too small, a corrupt stream, NULL blocks, a bad level)
// are mixed in and must fail with the right code while the other callers' calls go on undisturbed; the CPU budget is
// changed while calls run.  Prints one JSON line.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dswx_codec.h"

static std::atomic<int> g_failures{0};
static std::atomic<long long> g_calls{0}, g_blocks{0}, g_expected_errors{0};

#define CHECK(cond, what)                                                        \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "FAILED %s (%s:%d): %s\n", what, __FILE__, __LINE__, dswx_codec_last_error()); \
            g_failures.fetch_add(1);                                             \
            return;                                                              \
        }                                                                        \
    } while (0)

// An LZW ENCODER of the harness's own (TIFF 6.0 section 13 as libtiff writes it: most significant bit first, clear code at
// the start and when entry 4094 is due, the width grows when the encoder's next entry passes 2^width - 1), so that the
// library's decoder is checked against an independent statement of the format, not only against files from Pillow.
static std::vector<unsigned char> lzw_encode(const std::vector<unsigned char>& in) {
    std::vector<unsigned char> out;
    uint64_t acc = 0;
    int have = 0, nbits = 9, next = 258;
    auto put = [&](int code) {
        acc = (acc << nbits) | (uint64_t)code;
        have += nbits;
        while (have >= 8) { out.push_back((unsigned char)(acc >> (have - 8))); have -= 8; }
    };
    std::unordered_map<uint32_t, int> table;
    put(256);
    int w = -1;
    for (unsigned char c : in) {
        if (w < 0) { w = c; continue; }
        const uint32_t key = ((uint32_t)w << 8) | c;
        auto it = table.find(key);
        if (it != table.end()) { w = it->second; continue; }
        put(w);
        table[key] = next++;
        if (next == 4094) { put(256); table.clear(); next = 258; nbits = 9; }
        else if (next > (1 << nbits) - 1) ++nbits;
        w = c;
    }
    if (w >= 0) {
        put(w);
        ++next;
        if (next > (1 << nbits) - 1 && nbits < 12) ++nbits;
    }
    put(257);
    if (have) out.push_back((unsigned char)(acc << (8 - have)));
    return out;
}

static void lzw_caller(int id, int rounds) {
    std::mt19937_64 rng(777 + id);
    for (int r = 0; r < rounds; ++r) {
        const int n = 1 + (int)(rng() % 40);
        const int threads = 1 + (int)(rng() % 12);
        std::vector<std::vector<unsigned char>> src(n), enc(n), dec(n);
        std::vector<const void*> ep(n);
        std::vector<void*> dp(n);
        std::vector<size_t> es(n), dc(n), ds(n);
        for (int i = 0; i < n; ++i) {
            const size_t len = (rng() % 11 == 0) ? 0 : 1 + (size_t)(rng() % 300000);
            src[i].resize(len);
            const int kind = (int)(rng() % 4);
            for (size_t k = 0; k < len; ++k)
                src[i][k] = kind == 0 ? (unsigned char)(rng() & 0xff) : kind == 1 ? (unsigned char)((k / 97) % 5) : kind == 2 ? 0 : (unsigned char)((k % 7) * (k % 3));
            enc[i] = lzw_encode(src[i]);
            dec[i].assign(len + 1, 0xEE);
            ep[i] = enc[i].data(); es[i] = enc[i].size();
            dp[i] = dec[i].data(); dc[i] = dec[i].size();
        }
        CHECK(dswx_codec_unlzw_blocks(ep.data(), es.data(), dp.data(), dc.data(), ds.data(), n, threads) == DSWX_CODEC_OK, "unlzw");
        for (int i = 0; i < n; ++i)
            CHECK(ds[i] == src[i].size() && (ds[i] == 0 || memcmp(dec[i].data(), src[i].data(), ds[i]) == 0) && dec[i][ds[i]] == 0xEE, "LZW round trip");
        g_calls.fetch_add(1);
        g_blocks.fetch_add(n);
        // a destination shorter than the stream: filled, no error, nothing written behind it (libtiff's behaviour)
        for (int i = 0; i < n; ++i) if (src[i].size() > 10) {
            dc[i] = src[i].size() / 2;
            dec[i].assign(src[i].size() + 1, 0xEE);
            dp[i] = dec[i].data();
        }
        CHECK(dswx_codec_unlzw_blocks(ep.data(), es.data(), dp.data(), dc.data(), ds.data(), n, threads) == DSWX_CODEC_OK, "unlzw into short blocks");
        for (int i = 0; i < n; ++i) if (src[i].size() > 10)
            CHECK(ds[i] == dc[i] && memcmp(dec[i].data(), src[i].data(), ds[i]) == 0 && dec[i][ds[i]] == 0xEE, "short LZW block");
        // damaged streams: any outcome but a memory error (the sanitizers watch): OK or a data error
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 4 && !enc[i].empty(); ++k) enc[i][rng() % enc[i].size()] ^= (unsigned char)(1 + rng() % 255);
            if (rng() % 3 == 0) enc[i].resize(enc[i].size() / 2);
            ep[i] = enc[i].data(); es[i] = enc[i].size();
            if (es[i] == 0) { enc[i].push_back(0x80); ep[i] = enc[i].data(); es[i] = 1; }
            dc[i] = src[i].size() / 2 + 1;
            dec[i].assign(dc[i] + 1, 0xEE);
            dp[i] = dec[i].data();
        }
        const int rc = dswx_codec_unlzw_blocks(ep.data(), es.data(), dp.data(), dc.data(), ds.data(), n, threads);
        CHECK(rc == DSWX_CODEC_OK || rc == DSWX_CODEC_ERR_DATA, "damaged LZW stream");
        for (int i = 0; i < n; ++i) CHECK(dec[i][dc[i]] == 0xEE, "damaged LZW stream wrote behind its block");
        g_expected_errors.fetch_add(rc != DSWX_CODEC_OK);
    }
}

static void caller(int id, int rounds) {
    std::mt19937_64 rng(1234 + id);
    for (int r = 0; r < rounds; ++r) {
        const int n = 1 + (int)(rng() % 97);
        const int threads = 1 + (int)(rng() % 12);
        const int level = 1 + (int)(rng() % 9);
        std::vector<std::vector<unsigned char>> src(n), enc(n), dec(n);
        std::vector<const void*> sp(n);
        std::vector<void*> ep(n), dp(n);
        std::vector<size_t> ss(n), ec(n), es(n), dc(n), ds(n);
        for (int i = 0; i < n; ++i) {
            const size_t len = (rng() % 11 == 0) ? 0 : 1 + (size_t)(rng() % 70000);
            src[i].resize(len);
            const int kind = (int)(rng() % 3);
            for (size_t k = 0; k < len; ++k) src[i][k] = kind == 0 ? (unsigned char)(rng() & 0xff) : kind == 1 ? (unsigned char)((k / 97) % 5) : 0;
            enc[i].resize(dswx_codec_deflate_bound(len));
            dec[i].resize(len + 1);
            sp[i] = src[i].data(); ss[i] = len;
            ep[i] = enc[i].data(); ec[i] = enc[i].size();
            dp[i] = dec[i].data(); dc[i] = dec[i].size();
        }
        CHECK(dswx_codec_deflate_blocks(sp.data(), ss.data(), ep.data(), ec.data(), es.data(), n, level, threads) == DSWX_CODEC_OK, "deflate");
        std::vector<const void*> ecp(ep.begin(), ep.end());
        CHECK(dswx_codec_inflate_blocks(ecp.data(), es.data(), dp.data(), dc.data(), ds.data(), n, threads) == DSWX_CODEC_OK, "inflate");
        for (int i = 0; i < n; ++i)
            CHECK(ds[i] == ss[i] && (ss[i] == 0 || memcmp(dec[i].data(), src[i].data(), ss[i]) == 0), "round trip");
        g_calls.fetch_add(2);
        g_blocks.fetch_add(2 * n);
        // calls that must fail, while the other callers' calls run
        const int which = (int)(rng() % 5);
        if (which == 0 && n > 1) {                                   // a destination too small for its stream
            int big = -1;
            for (int i = 0; i < n; ++i) if (ss[i] > 64) big = i;
            if (big >= 0) {
                dc[big] = ss[big] - 1;
                CHECK(dswx_codec_inflate_blocks(ecp.data(), es.data(), dp.data(), dc.data(), ds.data(), n, threads) == DSWX_CODEC_ERR_SPACE, "too small a destination");
                g_expected_errors.fetch_add(1);
            }
        } else if (which == 1) {                                     // a corrupt stream
            int big = -1;
            for (int i = 0; i < n; ++i) if (es[i] > 16) big = i;
            if (big >= 0) {
                enc[big][es[big] / 2] ^= 0x5a;
                enc[big][es[big] - 1] ^= 0xff;                       // (the checksum too)
                const int rc = dswx_codec_inflate_blocks(ecp.data(), es.data(), dp.data(), dc.data(), ds.data(), n, threads);
                CHECK(rc == DSWX_CODEC_ERR_DATA || rc == DSWX_CODEC_ERR_SPACE, "corrupt stream");
                g_expected_errors.fetch_add(1);
            }
        } else if (which == 2) {
            CHECK(dswx_codec_deflate_blocks(sp.data(), ss.data(), ep.data(), ec.data(), es.data(), n, 0, threads) == DSWX_CODEC_ERR_ARG, "level 0");
            CHECK(dswx_codec_deflate_blocks(nullptr, ss.data(), ep.data(), ec.data(), es.data(), n, 6, threads) == DSWX_CODEC_ERR_ARG, "NULL table");
            g_expected_errors.fetch_add(2);
        } else if (which == 3) {
            dswx_codec_set_cpu_budget(1 + (int)(rng() % 6));         // the budget moves while calls run
        } else {
            dswx_codec_set_cpu_budget(0);
        }
    }
}

int main(int argc, char** argv) {
    const int callers = argc > 1 ? atoi(argv[1]) : 6, rounds = argc > 2 ? atoi(argv[2]) : 40;
    for (int engine = 0; engine < 2; ++engine) {
        dswx_codec_force_zlib(engine);
        std::vector<std::thread> ts;
        for (int c = 0; c < callers; ++c) ts.emplace_back(caller, 100 * engine + c, rounds);
        ts.emplace_back(lzw_caller, engine, rounds / 2 + 1);          // the LZW decoder shares the pool with them
        for (auto& t : ts) t.join();
    }
    dswx_codec_force_zlib(0);
    printf("{\"failures\": %d, \"calls\": %lld, \"blocks\": %lld, \"expected_errors\": %lld, \"engine\": \"%s\", \"cpu_budget\": %d}\n",
           g_failures.load(), g_calls.load(), g_blocks.load(), g_expected_errors.load(), dswx_codec_engine(), dswx_codec_cpu_budget());
    return g_failures.load() ? 1 : 0;
}
