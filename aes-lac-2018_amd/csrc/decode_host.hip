// Host-side helpers of the decode / scoring path: no device memory, no stream.
//   ds2_edit_distance   -- the Levenshtein distance Decoder.wer / Decoder.cer are built on (the reference imports the
//                          python-Levenshtein package, codes/decoder.py:20,49-78)
//   ds2_ctc_beam_search -- CTC prefix beam search over one utterance.  NOT in the reference (test.py:21 offers only
//                          greedy / none); listed as the last "next" row of SURVEY.md section 8f.  Validated against
//                          exhaustive enumeration on small cases and against the greedy decoder on peaked inputs.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "ds2_host.h"

extern "C" int ds2_edit_distance(const int32_t* a, int na, const int32_t* b, int nb) {
    DS2_CHECK_ARG(na >= 0 && nb >= 0 && (a || na == 0) && (b || nb == 0));
    if (na < nb) {
        std::swap(a, b);
        std::swap(na, nb);
    }
    std::vector<int> row(nb + 1);
    for (int j = 0; j <= nb; ++j) row[j] = j;
    for (int i = 1; i <= na; ++i) {
        int diag = row[0];   // D[i-1][j-1]
        row[0] = i;
        for (int j = 1; j <= nb; ++j) {
            const int up = row[j];
            const int best = std::min(std::min(up, row[j - 1]) + 1, diag + (a[i - 1] != b[j - 1]));
            diag = up;
            row[j] = best;
        }
    }
    return row[nb];
}

namespace {

constexpr double NEG_INF = -1e300;

inline double log_add(double x, double y) {
    if (x <= NEG_INF) return y;
    if (y <= NEG_INF) return x;
    const double m = x > y ? x : y;
    return m + log1p(exp(-fabs(x - y)));
}

struct Prefix {          // node of the prefix trie
    int parent, sym, born;   // born = frame at which the prefix first entered a beam (reported as the offset)
};
struct Score {
    double pb = NEG_INF, pnb = NEG_INF;   // log p(prefix, last frame blank / not blank)
    double total() const { return log_add(pb, pnb); }
};

}  // namespace

// probs: (T, A) row-major host array of probabilities (log_input = 0) or log-probabilities (log_input = 1).
// Keeps the beam_width most probable PREFIXES (alignments that collapse to the same labelling are merged) after
// every frame; returns the best one: its labels, the frame each label first appeared at, and log p(labelling).
extern "C" int ds2_ctc_beam_search(const float* probs, int T, int A, int blank, int beam_width, int log_input,
                                   int32_t* out_labels, int32_t* out_offsets, int out_cap, int* out_len,
                                   float* out_logp) {
    DS2_CHECK_ARG(probs && out_len && T >= 0 && A > 0 && blank >= 0 && blank < A && beam_width > 0);
    DS2_CHECK_ARG(out_cap >= 0 && (out_labels || out_cap == 0));
    std::vector<Prefix> trie(1, Prefix{-1, -1, 0});                 // node 0 = empty prefix
    std::unordered_map<long long, int> child;                       // (node << 20 | sym) -> node
    auto extend = [&](int node, int sym, int t) {
        const long long key = ((long long)node << 20) | (long long)sym;
        auto it = child.find(key);
        if (it != child.end()) return it->second;
        trie.push_back(Prefix{node, sym, t});
        child.emplace(key, (int)trie.size() - 1);
        return (int)trie.size() - 1;
    };
    DS2_CHECK_ARG(A < (1 << 20));
    std::vector<std::pair<int, Score>> beam(1);
    beam[0].first = 0;
    beam[0].second.pb = 0.0;
    std::unordered_map<int, Score> next;
    std::vector<double> lp(A);
    for (int t = 0; t < T; ++t) {
        for (int c = 0; c < A; ++c) {
            const double v = probs[(size_t)t * A + c];
            lp[c] = log_input ? v : (v > 0.0 ? log(v) : NEG_INF);
        }
        next.clear();
        for (const auto& kv : beam) {
            const int node = kv.first;
            const Score& s = kv.second;
            const double tot = s.total();
            const int last = trie[node].sym;
            {   // stay on the same prefix: a blank, or a repeat of its last symbol
                Score& same = next[node];
                same.pb = log_add(same.pb, tot + lp[blank]);
                if (last >= 0) same.pnb = log_add(same.pnb, s.pnb + lp[last]);
            }
            for (int c = 0; c < A; ++c) {
                if (c == blank || lp[c] <= NEG_INF) continue;
                // extending by the last symbol again needs a blank in between
                const double from = (c == last) ? s.pb : tot;
                if (from <= NEG_INF) continue;
                Score& ext = next[extend(node, c, t)];
                ext.pnb = log_add(ext.pnb, from + lp[c]);
            }
        }
        beam.assign(next.begin(), next.end());
        const size_t keep = std::min<size_t>(beam.size(), (size_t)beam_width);
        std::partial_sort(beam.begin(), beam.begin() + keep, beam.end(),
                          [](const std::pair<int, Score>& x, const std::pair<int, Score>& y) {
                              const double a = x.second.total(), b = y.second.total();
                              return a > b || (a == b && x.first < y.first);
                          });
        beam.resize(keep);
    }
    int best = 0;
    double best_lp = NEG_INF;
    for (const auto& kv : beam) {
        const double v = kv.second.total();
        if (v > best_lp || (v == best_lp && kv.first < best)) {
            best_lp = v;
            best = kv.first;
        }
    }
    std::vector<int> labels, offs;
    for (int n = best; n > 0; n = trie[n].parent) {
        labels.push_back(trie[n].sym);
        offs.push_back(trie[n].born);
    }
    const int len = (int)labels.size();
    *out_len = len;
    if (out_logp) *out_logp = (float)best_lp;
    if (len > out_cap) {
        ds2_set_error("ds2_ctc_beam_search: output of %d labels does not fit out_cap=%d", len, out_cap);
        return DS2_ERR_ARG;
    }
    for (int i = 0; i < len; ++i) {
        out_labels[i] = labels[len - 1 - i];
        if (out_offsets) out_offsets[i] = offs[len - 1 - i];
    }
    return DS2_OK;
}
