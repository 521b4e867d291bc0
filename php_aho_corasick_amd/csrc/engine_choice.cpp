// engine_choice.cpp -- automatic mode: the measured choice of the scan strategy
#include "engine.hpp"

namespace acmi_engine {

// ACMI_CHOICE_TRACE=1: every decision of the automatic mode on stderr (tools/exp_choice.py reads it)
static bool choice_trace() {
    static const bool on = std::getenv("ACMI_CHOICE_TRACE") != nullptr;
    return on;
}

// ---- automatic mode: the measured choice of the scan strategy --------------------------------------------------------------
// Batches are told apart by size (small ones are launch-bound whatever scans them) and by whether their haystacks are short
// (under 512 B on average: the anchor kernel spends a wave step per work item, the walks a lane): what one kind of batch
// taught the handle says little about the other.
int choice_class(uint64_t blob_bytes, size_t nh) {
    const int size = blob_bytes < (16ull << 20) ? 0 : blob_bytes < (256ull << 20) ? 1 : blob_bytes < (4ull << 30) ? 2 : 3;
    return size | ((nh && blob_bytes / nh < 512) ? 4 : 0);
}

// What a GiB is EXPECTED to cost (ms) on a strategy this handle has not run yet: the rate the kernel reaches on text it
// suits.  Only decides whether an untried alternative is worth a probe (one whose best case is no better than what the current
// strategy measures is never tried); every decision between strategies that have run is taken on measured times.
float nominal_cost(const acmi_handle *h, int strat, uint64_t blob_bytes, size_t nh, double density) {
    const FlatAutomaton &fa = h->aut.flat();
    const double avg = nh ? double(blob_bytes) / double(nh) : 0.0;
    // `density` = records per byte of the batch just ended: what the text is like is known even for a strategy that never ran on
    // it -- the filter pays for every candidate (a match is one), the optimistic pass walks again wherever there is an event,
    // every record is 16 bytes written (17 GB per GiB of text at a record per byte).  Without these terms a^1..a^16 kept
    // probing kernels that had no chance (each probe a batch at a third of the rate).
    switch (strat) {
    case acmi_handle::kStratAnchor: {
        double c = fa.anchor_dense ? 0.45 : fa.anchor_mode == 2 ? 0.30 : 0.19;
        if (avg < 2048.0) c *= 2048.0 / std::max(avg, 128.0) * 0.5 + 0.5;          // (a wave step per work item: short haystacks idle its lanes)
        return float(c + density * 250.0);
    }
    case acmi_handle::kStratLds: return float((fa.compact ? 2.0 : 0.33) + density * 300.0);
    case acmi_handle::kStratTiles: {
        // (the whole automaton in LDS, chain-compressed: round 6's walk costs 0.68-0.74 ms per GiB whatever the text, the call 0.78 at
        // a record per 128 bytes and 0.98 at one per 32; the logs turn into records at ~ 6 us per million.  Round 5's 1.0 kept the
        // headline set at 64 occurrences per 8 KiB on the filter's 0.90-0.92 ms: 1.2 x the expectation was above it, no probe -- the
        // expectation is the walk's own time, the best case, so that the margin does not hide a 15 % better strategy)
        if (h->chain_walk != 0 && chain_walk_ok(h->dA) && (h->chain_walk > 0 || h->dA.chain_dense < h->dA.chain_index)) return float(0.62 + density * 9.0);
        const bool allhot = uint64_t(fa.trap) * (fa.n_columns * fa.entry_bytes + 4u) <= 60u * 1024u && fa.trap + 1u == fa.n_states;
        return float((fa.compact ? 6.0 : allhot ? 1.0 : 1.45) + density * (density > 0.25 ? 20.0 : 4.5));   // (past an event per 4 bytes: the second walk)
    }
    default: return float((fa.compact ? 8.0 : 1.1) + density * 40.0);   // (measured: 1.43 / 2.37 ms per GiB at a record per 128 / 32 bytes)
    }
}

bool strategy_eligible(const acmi_handle *h, int strat, uint64_t blob_bytes, size_t nh, int find_all) {
    const FlatAutomaton &fa = h->aut.flat();
    uint32_t w = 0, p = 0;
    switch (strat) {
    case acmi_handle::kStratAnchor: return fa.anchor_q != 0;
    case acmi_handle::kStratTiles: return find_all && tiles_geometry(fa.max_pattern_length, fa.n_states, blob_bytes, nh, &w, &p);
    default: return true;
    }
}

// Where a handle starts, before anything is measured: what the pattern set suggests.
int seed_strategy(const acmi_handle *h, uint64_t blob_bytes, size_t nh, int find_all) {
    const FlatAutomaton &fa = h->aut.flat();
    // text over the patterns' own alphabet would flag more of an 8 KiB item than its run list holds (keyword lists, dense sets):
    // the optimistic LDS-tile pass ends up walking whole items anyway
    const bool flagged = fa.est_flag_rate * 8192.0 > double(h->run_cap) && !h->is_part;
    if (fa.anchor_q && !(blob_bytes >= kAutoMinBytes && nh && blob_bytes / nh < kSmallItemBytes && !flagged)) return acmi_handle::kStratAnchor;
    if (!flagged) return acmi_handle::kStratLds;
    // the wave drain writes output denser than a record per two bytes as whole rows; everything else in one pass
    if (!h->dense_output && strategy_eligible(h, acmi_handle::kStratTiles, blob_bytes, nh, find_all)) return acmi_handle::kStratTiles;
    return acmi_handle::kStratItems;
}

int choose_strategy(acmi_handle *h, acmi_handle::FlySlot &F, uint64_t blob_bytes, size_t nh, int find_all) {
    acmi_handle::ChoiceClass &C = h->choice[find_all ? 1 : 0][F.choice_class];
    F.probe = false;
    if (C.cur < 0 || !strategy_eligible(h, C.cur, blob_bytes, nh, find_all)) C.cur = seed_strategy(h, blob_bytes, nh, find_all);
    if (blob_bytes < kAutoMinBytes) return C.cur;   // (launch-bound either way: decides nothing, probes nothing)
    for (int s = 0; s < acmi_handle::kStratCount; ++s)
        if (C.wait[s]) --C.wait[s];
    // of the alternatives waiting for a probe, the one that promises most first (what it is expected to cost on text of the density
    // the last batch had): the headline set at 256 matches per 8 KiB went filter (18 ms per GiB) -> tile walk of the LDS rows (9 ms,
    // three batches until one was timed) -> whole automaton in LDS (1.3 ms), ten batches on the way
    int best = -1;
    float best_cost = 0.f;
    for (int s = 0; s < acmi_handle::kStratCount; ++s) {
        if (!(C.pending >> s & 1u)) continue;
        if (s == C.cur || !strategy_eligible(h, s, blob_bytes, nh, find_all)) { C.pending &= ~(1u << s); continue; }
        const float c = nominal_cost(h, s, blob_bytes, nh, C.density);
        if (best < 0 || c < best_cost) { best = s; best_cost = c; }
    }
    if (best >= 0) {
        C.pending &= ~(1u << best);
        F.probe = true;
        ++h->probes;
        if (choice_trace()) std::fprintf(stderr, "[acmi choice] class %d: probe %d (in use %d)\n", F.choice_class, best, C.cur);
        return best;
    }
    return C.cur;
}

// An untried (or backed-off) alternative is probed when the strategy in use costs more than this times what the alternative is
// expected to: below the 1.3 x the cliff test allows (tests/test_gpu_cliffs.py) -- at 1.5 a 2,048-word list with 8 matches per
// 8 KiB sat at 2.93 ms per GiB on the filter, 1.49 x the whole-automaton walk's expected 1.97 (measured: 1.92), and whether it was
// ever probed was a matter of the third digit.
constexpr float kProbeMargin = 1.2f;

void learn_strategy(acmi_handle *h, const acmi_handle::FlySlot &F, double ms_per_gib, double density) {
    acmi_handle::ChoiceClass &C = h->choice[F.find_all ? 1 : 0][F.choice_class];
    const int s = F.strategy;
    if (choice_trace())
        std::fprintf(stderr, "[acmi choice] class %d: strategy %d%s measured %.3f ms/GiB (had %.3f), in use %d at %.3f, density %.5f\n", F.choice_class, s,
                     F.probe ? " (probe)" : "", ms_per_gib, double(C.cost[s]), C.cur, C.cur >= 0 ? double(C.cost[C.cur]) : 0.0, density);
    // What a strategy costs is the LOWER of its last two batches: a measurement errs upwards only (the two events span the host's
    // enqueue -- a descheduled host thread, a clock that has not ramped yet, another process on the chip), and one batch at 3.6 x
    // its usual time used to send the next three batches out as probes of every alternative (round 4's "cliff" of cfg3 / 256 B in
    // one bench run of two: the timed calls were the spike and two probes).  A probe stands on its own measurement: what the
    // strategy cost when it last ran was another text.
    const float now = float(ms_per_gib);
    C.density = float(density);
    const bool first = C.last[s] <= 0.f || (F.probe && s != C.cur);
    C.cost[s] = first ? now : std::min(C.last[s], now);
    C.last[s] = now;
    if (F.probe && s != C.cur) {
        // a probe wins when it is at least 1.12 x cheaper than what the strategy in use costs now (both MEASURED on this text: the
        // margin only has to cover the run-to-run spread, a few per cent; at 1.25 a 2,048-word list over prose stayed on a filter
        // that cost 2.77 ms per GiB next to the tile walk's 2.33); a lost one backs off
        if (C.cost[C.cur] > 0.f && C.cost[s] * 1.12f < C.cost[C.cur]) {
            C.wait[C.cur] = C.span[C.cur];   // (what has just been measured and beaten is not probed again at once)
            C.cur = s;
            // (... and what was waiting for a probe because the OLD strategy was so slow has to promise something against the new one)
            for (int a = 0; a < acmi_handle::kStratCount; ++a)
                if ((C.pending >> a & 1u) && C.cost[s] <= kProbeMargin * (nominal_cost(h, a, F.blob_bytes, F.nh, density) + 0.06f * float(1ull << 30) / float(F.blob_bytes)))
                    C.pending &= ~(1u << a);
            C.span[s] = 16;
            C.seen = 1;
            ++h->switches;
        } else if (C.span[s] == 16 && C.cost[s] < 1.5f * C.cost[C.cur]) {
            // a close first loss is measured once more at once: a strategy's first batch on a handle carries its kernel's first
            // launch (code object, LDS attribute) -- the 2,048-word list over prose lost 2.33 against 2.77 ms that way
            C.cost[s] = C.last[s] = 0.f;
            C.span[s] = 32;
            C.pending |= 1u << s;
        } else {
            C.wait[s] = C.span[s];
            C.span[s] = std::min<uint32_t>(C.span[s] * 2, 1024);
        }
        return;
    }
    if (s != C.cur) return;
    // The strategy in use: every alternative whose kernel, on text that suits it, would be clearly cheaper than this is probed --
    // once its back-off allows: what it cost on EARLIER text (a filter that lost on dense matches) does not keep it out for good,
    // the text may have changed.  (Expected cost: the kernel's rate plus what any pipeline costs whatever it scans, ~ 60 us --
    // small batches are launch-bound on every strategy and probe nothing.)
    // (... and only on the evidence of two batches in a row -- `cost` is then the lower of the two -- unless the one batch cost four
    // times the alternative: no spike explains that, and a filter on text full of matches is 18 ms per GiB a batch)
    if (C.seen < 2) ++C.seen;
    for (int a = 0; a < acmi_handle::kStratCount; ++a) {
        if (a == s || C.wait[a] || !strategy_eligible(h, a, F.blob_bytes, F.nh, F.find_all)) continue;
        const float ref = nominal_cost(h, a, F.blob_bytes, F.nh, density) + 0.06f * float(1ull << 30) / float(F.blob_bytes);
        if (C.cost[s] > (C.seen < 2 ? 4.0f : kProbeMargin) * ref) C.pending |= 1u << a;
    }
}

}  // namespace acmi_engine
