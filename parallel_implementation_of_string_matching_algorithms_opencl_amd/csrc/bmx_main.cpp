// bmx_main.cpp -- C++ host driver over libbmx.so.
//
// Keeps the I/O contract of the reference's console program
// (BoyreMoore/BoyreMoore/BoyreMoore.cpp): a text file and a pattern file in
// (defaults are the reference's hard-coded names, :77 and :82), the shift
// tables built on the host (:150-190), the search repeated 10 times with the
// text already on the device and the mean time printed (:211, :258, :288-292,
// :314-315), per-range hit counts printed (:294-295).  What it does NOT keep:
// the echo of the whole text (:92), the lossy 2-way split at spaces (:94-141)
// as the default partition, and the per-iteration context/JIT (:217-256).
//
//   bmx_cli [--text F] [--pattern F] [--iters N] [--positions] [--max-print K]
//           [--ranges P]      reference-compatible mode: split at spaces into P
//                             inclusive ranges like BoyreMoore.cpp:94-141 and
//                             print the per-range counts of bmx_search_ranges
//           [--device D]
//   bmx_cli --edit-distance A B [--iters N]   the reference's second program (EditDistance-1.cpp:
//                             two strings from files -- it opens str1.txt twice, :94-95 -- the mean time
//                             of the runs and the distance, :358-383)
//   bmx_cli --suffix-array F [--iters N] [--max-print K]   its third (SuffixArrays.cpp: text from
//                             input.txt, :181; array printed, :155-161; mean time, :514)
//           [--gpus G]        also run the search over G GPUs from this one process: devices, RCCL
//                             communicators and the text set up once (bmx_multi_*), `iters` searches on
//                             the resident shards, each list checked against the one-GPU list; then once
//                             through the host-buffer entry point bmx_search_multi
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "bmx.h"

namespace {

bool read_file(const std::string &path, std::string &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

// The reference's partition (BoyreMoore.cpp:94-141): words are maximal runs
// between single spaces; P ranges of numberOfWords / P words each, inclusive
// [start, end] with the separating space excluded; leftover words are dropped.
std::vector<int32_t> split_like_reference(const std::string &text, int P)
{
    std::vector<int32_t> word_len;
    int32_t cur = 0;
    for (char ch : text) {
        if (ch == ' ') {
            word_len.push_back(cur);
            cur = 0;
        } else {
            ++cur;
        }
    }
    word_len.push_back(cur);
    const size_t per = word_len.size() / (size_t)P;
    std::vector<int32_t> se;
    int64_t pos = 0;
    size_t w = 0;
    for (int r = 0; r < P; ++r) {
        int64_t start = pos, end = pos;
        for (size_t j = 0; j < per; ++j, ++w) end += word_len[w] + 1;
        se.push_back((int32_t)start);
        se.push_back((int32_t)(end - 2)); // last character of the last word
        pos = end;
    }
    return se;
}

} // namespace

int main(int argc, char **argv)
{
    std::string text_path = "inputEd.txt", pat_path = "input1Search.txt", ed_a, ed_b, sa_path;
    int iters = 10, device = 0, ranges = 0, gpus = 0;
    bool positions = false;
    uint64_t max_print = 32;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "%s needs a value\n", name);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--text") text_path = need("--text");
        else if (a == "--pattern") pat_path = need("--pattern");
        else if (a == "--iters") iters = atoi(need("--iters"));
        else if (a == "--device") device = atoi(need("--device"));
        else if (a == "--gpus") gpus = atoi(need("--gpus"));
        else if (a == "--ranges") ranges = atoi(need("--ranges"));
        else if (a == "--max-print") max_print = strtoull(need("--max-print"), nullptr, 10);
        else if (a == "--positions") positions = true;
        else if (a == "--edit-distance") {
            ed_a = need("--edit-distance");
            ed_b = need("--edit-distance");
        } else if (a == "--suffix-array") sa_path = need("--suffix-array");
        else {
            fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }

    if (!ed_a.empty() || !sa_path.empty()) {
        bmx_ctx *ctx = nullptr;
        int rc = bmx_ctx_create(device, &ctx);
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_ctx_create failed: %d (%s)\n", rc, bmx_last_error());
            return 1;
        }
        double total = 0.0;
        if (!ed_a.empty()) {
            std::string x, y;
            if (!read_file(ed_a, x) || !read_file(ed_b, y)) {
                fprintf(stderr, "File Not Found!\n"); // EditDistance-1.cpp:99
                return 1;
            }
            uint64_t d = 0;
            for (int it = 0; it < iters; ++it) {
                auto t0 = std::chrono::steady_clock::now();
                rc = bmx_edit_distance(ctx, x.data(), x.size(), y.data(), y.size(), &d);
                auto t1 = std::chrono::steady_clock::now();
                if (rc != BMX_OK) {
                    fprintf(stderr, "bmx_edit_distance failed: %d (%s)\n", rc, bmx_last_error());
                    return 1;
                }
                total += std::chrono::duration<double>(t1 - t0).count();
            }
            printf("%llu %llu\n", (unsigned long long)x.size(), (unsigned long long)y.size());
            printf("%llu\n", (unsigned long long)d);                                  // :369
            if (iters > 0) printf("\n\nAverage time :%f \n", total / iters);          // :383
        } else {
            std::string t;
            if (!read_file(sa_path, t)) {
                fprintf(stderr, "File Not Found!\n"); // SuffixArrays.cpp:185
                return 1;
            }
            std::vector<int32_t> sa(t.size() ? t.size() : 1);
            for (int it = 0; it < iters; ++it) {
                auto t0 = std::chrono::steady_clock::now();
                rc = bmx_suffix_array(ctx, t.data(), t.size(), sa.data());
                auto t1 = std::chrono::steady_clock::now();
                if (rc != BMX_OK) {
                    fprintf(stderr, "bmx_suffix_array failed: %d (%s)\n", rc, bmx_last_error());
                    return 1;
                }
                total += std::chrono::duration<double>(t1 - t0).count();
            }
            printf("%llu\n", (unsigned long long)t.size()); // :198
            for (uint64_t i = 0; i < t.size() && i < max_print; ++i) printf("%d ", sa[i]); // :158-160
            printf("\n");
            if (iters > 0) printf("Average Time  = %f\n", total / iters); // :514
        }
        bmx_ctx_destroy(ctx);
        return 0;
    }

    std::string text, pat;
    if (!read_file(text_path, text)) {
        fprintf(stderr, "cannot read text file %s\n", text_path.c_str());
        return 1;
    }
    if (!read_file(pat_path, pat)) {
        fprintf(stderr, "cannot read pattern file %s\n", pat_path.c_str());
        return 1;
    }
    const uint64_t n = text.size();
    const int32_t m = (int32_t)pat.size();
    printf("text %s: %llu bytes, pattern %s: %d bytes\n", text_path.c_str(), (unsigned long long)n,
           pat_path.c_str(), m);

    int32_t bad[BMX_BAD_TABLE_SIZE];
    std::vector<int32_t> good(m > 0 ? m : 1);
    int rc = bmx_build_tables(pat.data(), m, bad, good.data());
    if (rc != BMX_OK) {
        fprintf(stderr, "bmx_build_tables failed: %d\n", rc);
        return 1;
    }

    bmx_ctx *ctx = nullptr;
    rc = bmx_ctx_create(device, &ctx);
    if (rc != BMX_OK) {
        fprintf(stderr, "bmx_ctx_create failed: %d (%s)\n", rc, bmx_last_error());
        return 1;
    }

    if (ranges > 0) {
        std::vector<int32_t> se = split_like_reference(text, ranges);
        std::vector<int32_t> ans(ranges);
        rc = bmx_search_ranges(ctx, text.data(), n, pat.data(), se.data(), ranges, ans.data(), good.data(), bad, m);
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_search_ranges failed: %d (%s)\n", rc, bmx_last_error());
            return 1;
        }
        for (int r = 0; r < ranges; ++r)
            printf("The no. of occurrences by process %d is %d   [range %d..%d]\n", r, ans[r], se[2 * r], se[2 * r + 1]);
    }

    // text resident once, searched `iters` times (the reference's timer also
    // starts after the upload)
    void *d_text = nullptr;
    uint64_t *d_out = nullptr;
    const uint64_t cap = n >= (uint64_t)m ? n - (uint64_t)m + 1 : 1;
    rc = bmx_text_upload(ctx, text.data(), n, &d_text);
    if (rc == BMX_OK) rc = bmx_device_alloc(ctx, cap * sizeof(uint64_t), (void **)&d_out);
    if (rc != BMX_OK) {
        fprintf(stderr, "device setup failed: %d (%s)\n", rc, bmx_last_error());
        return 1;
    }
    double total = 0.0;
    uint64_t n_matches = 0;
    for (int it = 0; it < iters; ++it) {
        auto t0 = std::chrono::steady_clock::now();
        rc = bmx_search_device(ctx, d_text, n, n, 0, pat.data(), m, good.data(), bad, d_out, cap, &n_matches, nullptr);
        auto t1 = std::chrono::steady_clock::now();
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_search_device failed: %d (%s)\n", rc, bmx_last_error());
            return 1;
        }
        const double s = std::chrono::duration<double>(t1 - t0).count();
        total += s;
        printf("Time Spent: %.6f s (scan kernel %.3f ms)\n", s, bmx_last_scan_ms(ctx));
    }
    printf("occurrences: %llu\n", (unsigned long long)n_matches);
    if (iters > 0) {
        const double avg = total / iters;
        printf("Average time = %.6f s  (%.3f GB/s)\n", avg, avg > 0 ? (double)n / avg / 1e9 : 0.0);
    }

    if (gpus > 0) {
        // The same search from this ONE host process over `gpus` devices (the reference drives all of its work-items from
        // one main, BoyreMoore.cpp:273-286): devices, RCCL communicators and the text are set up once (bmx_multi_*), then
        // `iters` searches run on the resident shards, each ending with one all-gather of match-offset slots.
        std::vector<uint64_t> one(n_matches ? n_matches : 1), many(n_matches ? n_matches : 1);
        uint64_t got1 = 0, gotN = 0;
        rc = bmx_search(ctx, text.data(), n, pat.data(), m, one.data(), one.size(), &got1);
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_search failed: %d (%s)\n", rc, bmx_last_error());
            return 1;
        }
        bmx_multi *mg = nullptr;
        rc = bmx_multi_create(nullptr, gpus, &mg);
        if (rc == BMX_OK) rc = bmx_multi_text_upload(mg, text.data(), n, m);
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_multi over %d GPUs: set-up failed: %d (%s)\n", gpus, rc, bmx_last_error());
            return 1;
        }
        double total_multi = 0.0;
        for (int it = 0; it < iters; ++it) {
            auto t0 = std::chrono::steady_clock::now();
            rc = bmx_multi_search(mg, pat.data(), m, many.data(), many.size(), &gotN);
            auto t1 = std::chrono::steady_clock::now();
            if (rc != BMX_OK) {
                fprintf(stderr, "bmx_multi_search over %d GPUs failed: %d (%s)\n", gpus, rc, bmx_last_error());
                return 1;
            }
            total_multi += std::chrono::duration<double>(t1 - t0).count();
        }
        const char *how[] = {"none", "RCCL all-gather of slots", "slots staged through host memory", "exact path (dense result)"};
        bool same = got1 == gotN && std::equal(one.begin(), one.begin() + got1, many.begin());
        printf("%d GPUs, resident shards: %llu occurrences, average time = %.6f s (slowest scan kernel %.3f ms, exchange: %s), list %s the one-GPU list\n",
               gpus, (unsigned long long)gotN, iters > 0 ? total_multi / iters : 0.0, bmx_multi_last_scan_ms(mg),
               how[bmx_multi_last_exchange(mg) & 3], same ? "identical to" : "DIFFERS from");
        bmx_multi_destroy(mg);
        if (!same) return 1;
        // ... and through the host-buffer entry point (upload + search + download per call)
        auto t0 = std::chrono::steady_clock::now();
        rc = bmx_search_multi(text.data(), n, pat.data(), m, nullptr, gpus, many.data(), many.size(), &gotN);
        auto t1 = std::chrono::steady_clock::now();
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_search_multi over %d GPUs failed: %d (%s)\n", gpus, rc, bmx_last_error());
            return 1;
        }
        same = got1 == gotN && std::equal(one.begin(), one.begin() + got1, many.begin());
        printf("%d GPUs, host buffers in and out: %llu occurrences in %.6f s, list %s the one-GPU list\n", gpus,
               (unsigned long long)gotN, std::chrono::duration<double>(t1 - t0).count(),
               same ? "identical to" : "DIFFERS from");
        if (!same) return 1;
    }

    // the positions, through the host-buffer entry point (text, pattern, match_positions)
    if (positions) {
        std::vector<uint64_t> pos(n_matches ? n_matches : 1);
        uint64_t got = 0;
        rc = bmx_search(ctx, text.data(), n, pat.data(), m, pos.data(), pos.size(), &got);
        if (rc != BMX_OK) {
            fprintf(stderr, "bmx_search failed: %d (%s)\n", rc, bmx_last_error());
            return 1;
        }
        for (uint64_t i = 0; i < got && i < max_print; ++i) printf("Found at : %llu\n", (unsigned long long)pos[i]);
        if (got > max_print) printf("... %llu more\n", (unsigned long long)(got - max_print));
    }
    bmx_device_free(ctx, d_out);
    bmx_device_free(ctx, d_text);
    bmx_ctx_destroy(ctx);
    return 0;
}
