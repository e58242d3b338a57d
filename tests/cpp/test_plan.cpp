// CPU unit tests of the library's host-side planning (remotesensingproject_amd/csrc/rslf_plan.hpp): built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// by tests/test_plan_cpu.py.  No HIP, no GPU: the same header the .hip translation units include.
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <set>
#include <vector>

#include "rslf_plan.hpp"

using namespace rslf;
using namespace rslf::plan;

static int g_checks = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        g_checks++;                                                             \
        if (!(cond)) {                                                          \
            std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

static void test_sweep_order()
{
    // core.hpp:981-990: centre first, then +1, -1, +2, -2 ...: every view once for odd S; for even S the centre is S/2
    // and the loop ends before view 0 (the reference never visits it)
    for (int S = 1; S <= 64; S++) {
        const std::vector<int> o = sweep_order(S);
        const int mid = S / 2;
        CHECK(!o.empty() && o[0] == mid);
        std::set<int> seen(o.begin(), o.end());
        CHECK(seen.size() == o.size());
        for (int s : o)
            CHECK(s >= 0 && s < S);
        // the reference's loop: s_mid + off for off < S - s_mid, s_mid - off while it stays > -1
        const int expect = 1 + (S - mid - 1) + std::min(mid, S - mid - 1);
        CHECK((int)o.size() == expect);
        for (size_t i = 0; i + 1 < o.size(); i++)
            CHECK(sweep_view_after(S, o[i]) == o[i + 1]);
        CHECK(sweep_view_after(S, o.back()) == -1);
        CHECK(sweep_view_after(S, S + 5) == -1);
    }
    CHECK(sweep_order(0).empty());
    const std::vector<int> o5 = sweep_order(5);
    CHECK(o5.size() == 5 && o5[0] == 2 && o5[1] == 3 && o5[2] == 1 && o5[3] == 4 && o5[4] == 0);
    const std::vector<int> o4 = sweep_order(4);   // s_mid = 2: 2, 3, 1 -- view 0 is never visited
    CHECK(o4.size() == 3 && o4[0] == 2 && o4[1] == 3 && o4[2] == 1);
}

static void test_small_rules()
{
    CHECK(resolve_s_hat(-1, 9) == 4 && resolve_s_hat(9, 9) == 4 && resolve_s_hat(3, 9) == 3 && resolve_s_hat(0, 8) == 0);
    CHECK(resolve_s_hat(-1, 8) == 4 && resolve_s_hat(-1, 1) == 0);
    CHECK(mean_shift_passes(10.0f) == 10 && mean_shift_passes(0.5f) == 1 && mean_shift_passes(3.0001f) == 4);
    CHECK(mean_shift_passes(1e30f) == (1 << 20));
    int v2, u2;
    f2c_level_dims(5, 7, &v2, &u2);     // cvRound(2.5) = 2 (ties to even), cvRound(3.5) = 4
    CHECK(v2 == 2 && u2 == 4);
    f2c_level_dims(512, 1, &v2, &u2);   // cvRound(0.5) = 0
    CHECK(v2 == 256 && u2 == 0);
    CHECK(halo_rows(5, 1) == 2 && halo_rows(7, 1) == 3 && halo_rows(5, 3) == 4 && halo_rows(1, 1) == 0 && halo_rows(5, 31) == 32);
    CHECK(pick_spad(101, 1) == 104 && pick_spad(33, 1) == 40 && pick_spad(9, 1) == 16 && pick_spad(193, 1) == 0);
    CHECK(pick_spad(48, 3) == 48 && pick_spad(49, 3) == 0 && pick_spad(1, 3) == 8 && pick_spad(5, 2) == 0);
}

static void test_pyramid()
{
    std::vector<LevelDims> p = f2c_pyramid(512, 512, 0);
    CHECK(p.size() == 6 && p[0].V == 512 && p[5].V == 16 && p[5].U == 16);
    p = f2c_pyramid(512, 512, 2);
    CHECK(p.size() == 2 && p[1].V == 256);
    CHECK(f2c_pyramid(10, 512, 0).empty());
    CHECK(f2c_pyramid(11, 11, 0).size() == 1);
    p = f2c_pyramid(960, 540, 0);   // 960x540, 480x270, 240x135, 120x68, 60x34, 30x17, 15x8(stop: 8 <= 10)
    CHECK(p.size() == 6 && p[3].U == 68 && p[5].V == 30 && p[5].U == 17);
    for (int V = 1; V < 200; V += 7)
        for (int U = 1; U < 200; U += 11)
            for (const LevelDims& l : f2c_pyramid(V, U, 0))
                CHECK(l.V > kMinSpatialDim && l.U > kMinSpatialDim);
}

static void test_structuring_element()
{
    MorphElement e = structuring_element(0, 3);
    CHECK(e.k == 3 && e.rows[0] == 7u && e.rows[1] == 7u && e.rows[2] == 7u && e.rows[3] == 0u);
    e = structuring_element(1, 3);
    CHECK(e.rows[0] == 2u && e.rows[1] == 7u && e.rows[2] == 2u);
    e = structuring_element(2, 5);   // OpenCV 3.x ellipse 5x5: 00100 / 11111 / 11111 / 11111 / 00100
    CHECK(e.rows[0] == 4u && e.rows[1] == 31u && e.rows[2] == 31u && e.rows[3] == 31u && e.rows[4] == 4u);
    for (int shape = 0; shape < 3; shape++)
        for (int k = 1; k <= 31; k++) {
            e = structuring_element(shape, k);
            for (int i = 0; i < 31; i++) {
                if (i >= k)
                    CHECK(e.rows[i] == 0u);
                if (k < 31)
                    CHECK((e.rows[i] >> k) == 0u);
            }
            CHECK((e.rows[k / 2] >> (k / 2)) & 1u);   // the anchor belongs to every element
        }
}

static void check_chunks(const std::vector<RowBlock>& c, int r0, int r1, int V, int halo)
{
    if (r1 <= r0) {
        CHECK(c.empty());
        return;
    }
    CHECK(!c.empty() && c.front().a == r0 && c.back().b == r1);
    for (size_t i = 0; i < c.size(); i++) {
        CHECK(c[i].a < c[i].b);
        CHECK(c[i].lo == std::max(0, c[i].a - halo) && c[i].hi == std::min(V, c[i].b + halo));
        CHECK(c[i].lo >= 0 && c[i].hi <= V && c[i].lo <= c[i].a && c[i].hi >= c[i].b);
        if (i)
            CHECK(c[i].a == c[i - 1].b);
    }
    CHECK(max_held_rows(c) >= c.front().hi - c.front().lo);
}

static void test_partitions()
{
    std::mt19937 rng(20260003);
    for (int it = 0; it < 20000; it++) {
        const int V = 1 + (int)(rng() % 3000), n = 1 + (int)(rng() % 9), halo = (int)(rng() % 6);
        int covered = 0;
        for (int i = 0; i < n; i++) {
            const RowBlock b = row_block(V, i, n, halo);
            CHECK(b.a == covered && b.b >= b.a && b.lo >= 0 && b.hi <= V);
            CHECK(b.b - b.a == V / n || b.b - b.a == V / n + 1);
            covered = b.b;
            const int chunk_rows = (rng() % 3 == 0) ? 1 + (int)(rng() % 200) : 0;
            const bool scattered = rng() & 1;
            check_chunks(chunk_plan(b.a, b.b, V, halo, chunk_rows, scattered), b.a, b.b, V, halo);
        }
        CHECK(covered == V);
        const int nd = sweep_devices_for(V, n, halo);
        CHECK(nd >= 1 && nd <= n && (nd == 1 || V / nd >= std::max(1, halo)));
        if (nd < n)
            CHECK(V / (nd + 1) < std::max(1, halo));
    }
    // the c3 partition of BASELINE.json configs[3]: 8 x 135 rows, 2-row halos
    for (int i = 0; i < 8; i++) {
        const RowBlock b = row_block(1080, i, 8, 2);
        CHECK(b.b - b.a == 135 && b.hi - b.lo == (i == 0 || i == 7 ? 137 : 139));
    }
    // the automatic plan of the host path at c3: a short first chunk, then two large ones
    std::vector<RowBlock> c = chunk_plan(0, 1080, 1080, 2, 0, false);
    CHECK(c.size() == 3 && c[0].b - c[0].a == 68 && c[1].b - c[1].a == 506 && c[2].b - c[2].a == 506);
    c = chunk_plan(0, 1080, 1080, 2, 0, true);   // scattered EPIs: a middling second chunk, pieces of about V/3.5
    CHECK(c.size() == 5 && c[0].b - c[0].a == 68 && c[1].b - c[1].a == 135);
    CHECK(chunk_plan(5, 5, 10, 2, 0, false).empty());
    int a, b2;
    for (int rows = 1; rows < 50; rows++)
        for (int nt = 1; nt < 10; nt++) {
            int tot = 0;
            for (int t = 0; t < nt; t++) {
                split_range(rows, t, nt, &a, &b2);
                CHECK(a == tot && b2 >= a);
                tot = b2;
            }
            CHECK(tot == rows);
        }
}

static void test_host_copies()
{
    // EPIs of 3 rows x 8 bytes: stacked, then with a gap
    static char buf[1024];
    const size_t row = 8, epi = 24;
    const void* stacked[6];
    for (int i = 0; i < 6; i++)
        stacked[i] = buf + (size_t)i * epi;
    CHECK(count_runs(stacked, 6, row, row, epi) == 1);
    CHECK(!epis_scattered(stacked, 0, 6, row, row, epi));
    CHECK(count_runs(stacked, 6, 16, row, epi) == 6);          // a row stride: every EPI its own run
    CHECK(epis_scattered(stacked, 0, 6, 16, row, epi));
    const void* gap[6] = {buf, buf + 24, buf + 100, buf + 124, buf + 148, buf + 400};
    CHECK(count_runs(gap, 6, row, row, epi) == 3);
    CHECK(epis_scattered(gap, 0, 6, row, row, epi) && !epis_scattered(gap, 2, 5, row, row, epi));
    CHECK(count_runs(gap, 0, row, row, epi) == 0 && count_runs(gap, 1, row, row, epi) == 1);
    // a pinned buffer left by an earlier, smaller call must not be taken for a chunk it cannot hold
    CHECK(use_pinned_gather(9, 10, 100, 1000) && !use_pinned_gather(9, 11, 100, 1000) && !use_pinned_gather(8, 10, 100, 1000) &&
          !use_pinned_gather(100, 10, 100, 0));
    CHECK(staging_chunk_rows(1 << 20, 1080) == 256 && staging_chunk_rows((size_t)1 << 30, 1080) == 1 && staging_chunk_rows(100, 7) == 7);
    CHECK(staging_chunk_rows(0, 7) == 7);
}

static void test_plane_layout()
{
    for (int C : {1, 3})
        for (size_t n : {(size_t)1, (size_t)63, (size_t)1920 * 139}) {
            const PlaneLayout q = plane_layout(n, C, 139);
            CHECK(q.Ce == 0 && q.Cd == 4 * n && q.depth == 8 * n && q.raw == 12 * n && q.score == 16 * n && q.rbar == 20 * n);
            CHECK(q.idx == q.rbar + 4 * n * C && q.mask == q.idx + 4 * n && q.counts >= q.mask + n && q.counts % 16 == 0);
            CHECK(q.bytes == q.counts + 139 * sizeof(int));
        }
}

static ScanRequest request(int V, int U, int S, int C, int D)
{
    ScanRequest r{};
    r.V = V, r.U = U, r.S = S, r.C = C, r.dim_d = D;
    r.spad = pick_spad(S, C);
    r.use_stream = r.spad == 0;
    r.reg_waves = r.spad ? 3 : 0;
    r.num_cus = 256;
    r.ctx_groups = 1;
    r.ctx_packed = false;
    r.precompacted = 0;
    r.force_groups = 0;
    r.force_packed = -1;
    r.px_mode = -1;
    r.stream_groups = 0;
    r.stream_share = 1;
    r.stream_lds_bytes = kStreamLdsBytes;
    return r;
}

static void check_plan(const ScanRequest& r, const ScanPlan& p)
{
    CHECK(p.groups >= 1 && p.groups <= 64);
    CHECK(p.groups == 1 || r.dim_d >= 2 * kScanWavesPerTile * p.groups);   // every wave keeps at least two hypotheses
    CHECK(p.tile_w == 63 || p.tile_w == 64);
    CHECK((long long)p.tiles_per_row * 64 >= r.U);                          // the tiles cover a scanline
    CHECK(p.tile_w == 64 ? p.tiles_per_row == (r.U + 63) / 64 : (long long)(p.tiles_per_row - 1) * 63 + 64 >= r.U);
    CHECK(p.rows_per_launch >= 1 && p.rows_per_launch <= r.V);
    if (p.groups > 1) {
        CHECK(p.records > 0 && p.tickets > 0);
        if (!p.packed) {
            // a launch of rows_per_launch scanlines writes tiles * groups * 64 records and draws one ticket per tile
            const size_t tiles = (size_t)p.rows_per_launch * p.tiles_per_row;
            CHECK(p.records == tiles * p.groups * 64 && p.tickets == tiles);
            CHECK(p.records * kPartialRecordBytes <= kPartialBudget || p.rows_per_launch == 1);
        } else if (p.packed_adapt) {
            CHECK(p.records <= (size_t)kPackedItemTarget * 64 && p.tickets <= (size_t)kPackedItemTarget / 2);
        } else {
            CHECK(p.records * kPartialRecordBytes <= kPartialBudget || p.groups == 1);
        }
    } else {
        CHECK(p.records == 0 && p.tickets == 0);
    }
    if (r.use_stream) {
        CHECK(p.lds_bytes <= r.stream_lds_bytes || p.stream_park == 0);
        CHECK(p.stream_park >= 0 && p.stream_park % (r.C == 1 ? 8 : 4) == 0);
        CHECK((size_t)p.stream_wave_floats * sizeof(float) * kScanWavesPerTile == p.lds_bytes);
        CHECK((size_t)p.stream_wave_floats >= (((size_t)r.S + 3) & ~(size_t)3) + (size_t)p.stream_park * r.C * 64);
    } else {
        CHECK(p.lds_bytes == 0 && p.stream_park == 0);
    }
}

static void test_scan_plans()
{
    // the shapes DESIGN.md quotes
    ScanRequest c3 = request(1080, 1920, 101, 1, 256);
    ScanPlan p = plan_scan(c3, 0);
    CHECK(p.groups == 1 && !p.packed && p.tile_w == 64 && p.tiles_per_row == 30 && p.rows_per_launch == 1080 && p.records == 0);
    ScanRequest shard = request(139, 1920, 101, 1, 256);   // one GPU's share of eight: 5.4 rounds -> eight groups
    p = plan_scan(shard, 0);
    CHECK(p.groups == 8 && p.rows_per_launch == 139 && p.records == (size_t)139 * 30 * 8 * 64);
    CHECK(p.records * kPartialRecordBytes <= kAutoGroupBudget);
    ScanRequest c1 = request(960, 540, 9, 1, 64);           // 64 hypotheses over 9 views: too little work per wave for groups
    c1.reg_waves = 7;
    p = plan_scan(c1, 0);
    CHECK(p.groups == 1);
    ScanRequest c5 = request(2160, 4096, 201, 3, 512);      // streaming kernel: 16 groups, 63-pixel tiles, row blocks under the budget
    p = plan_scan(c5, 68);
    CHECK(p.groups == 16 && p.tile_w == 63 && p.tiles_per_row == 65 && p.rows_per_launch == 126 && p.stream_park == 24);
    CHECK(p.lds_bytes == 4 * 4 * (204 + 24 * 3 * 64) && p.lds_bytes <= kStreamLdsBytes);
    {   // a small EPI (MansionLR: 100 views x 1146 px RGB = 1.4 MB): three groups where 120 hypotheses divide over 12 waves, else two;
        // not with few tiles, not on request, not for the sparse rows' launches (which ask for kStreamGroups themselves)
        ScanRequest m = request(720, 1146, 100, 3, 120);
        m.spad = 0;
        m.use_stream = true;
        m.num_cus = 256;
        m.stream_lds_bytes = kStreamLdsBytes;
        CHECK(plan_scan(m, 68).groups == 3);
        m.dim_d = 128;
        CHECK(plan_scan(m, 68).groups == 2);
        m.V = 8;
        CHECK(plan_scan(m, 68).groups == 16);
        m.V = 720;
        m.stream_groups = kStreamGroups;
        CHECK(plan_scan(m, 68).groups == 16);
        m.stream_groups = 0;
        m.S = 201;
        m.U = 4096;
        CHECK(plan_scan(m, 68).groups == 16);
    }
    ScanRequest chip = c5;                                  // the on-chip kernel: the streaming kernel's tiles, half its groups, all of the CU's LDS
    chip.use_stream = false;
    chip.use_chip = true;
    chip.chip_wave_floats = chip_wave_floats(201, kChipLadder[chip_rung_for(201)]);
    CHECK(chip.chip_wave_floats == 204 + 50 * 3 * 64 + 256);
    p = plan_scan(chip, 0);
    CHECK(p.groups == kChipGroups && !p.packed && p.tile_w == 63 && p.tiles_per_row == 65 && p.rows_per_launch == 252 && p.stream_park == 0);
    chip.V = 3;                                             // a few scanlines: 3 x 65 tiles x 8 groups would be three rounds of workgroups on 256 CUs
    chip.num_cus = 256;
    CHECK(plan_scan(chip, 0).groups == kStreamGroups);
    chip.stream_groups = 4;                                 // the debug override holds
    CHECK(plan_scan(chip, 0).groups == 4);
    CHECK(p.lds_bytes == (size_t)4 * 4 * (204 + 50 * 3 * 64 + 256) && p.lds_bytes <= (size_t)160 << 10);
    // the sweep's sparse visits: a packed list, 32 groups asked for
    ScanRequest sparse = request(512, 512, 33, 1, 128);
    sparse.ctx_groups = kSweepGroups;
    sparse.ctx_packed = true;
    sparse.precompacted = 2;
    p = plan_scan(sparse, 0);
    // ... which the pixel-per-wave kernel does not need: 128 hypotheses = two waves of lanes per pixel, no groups, no records
    CHECK(p.packed && !p.packed_adapt && p.px_waves == 2 && p.groups == 1 && p.records == 0 && p.tickets == 0);
    sparse.px_mode = 0;                                     // switched off (debug key "px"): the pixel-per-lane kernel and its groups
    p = plan_scan(sparse, 0);
    CHECK(p.packed && p.packed_adapt && p.px_waves == 0 && p.groups == 16 && p.records == (size_t)kPackedItemTarget * 64);
    sparse.px_mode = -1;
    sparse.dim_d = 16;                                      // too few hypotheses to fill a wave's lanes: stays with the pixel-per-lane kernel
    CHECK(plan_scan(sparse, 0).px_waves == 0);
    sparse.px_mode = 1;                                     // ... unless forced (parity tests)
    CHECK(plan_scan(sparse, 0).px_waves == 1);
    sparse.px_mode = -1;
    sparse.dim_d = 128;
    CHECK(px_waves(256) == 4 && px_waves(128) == 2 && px_waves(120) == 2 && px_waves(64) == 1 && px_waves(40) == 1 && px_waves(512) == 4);
    CHECK(px_waves(32) == 0 && px_waves(16) == 0 && px_waves(2) == 0 && px_waves(300) == 1 && px_waves(200) == 4 && px_waves(100) == 2);
    {   // the streaming kernel's packed launches take the pixel-per-wave form too (k2_scan_stream_px, round 4): no groups, no
        // records, the same LDS split as the row kernel; switched off, they keep their groups
        ScanRequest many = request(64, 512, 250, 1, 128);
        many.ctx_packed = true;
        many.ctx_groups = 8;
        ScanPlan q = plan_scan(many, 192);
        CHECK(many.use_stream && q.px_waves == 2 && q.groups == 1 && q.records == 0 && !q.packed_adapt && q.lds_bytes > 0);
        many.px_mode = 0;
        q = plan_scan(many, 192);
        CHECK(q.px_waves == 0 && q.groups == 8 && q.records > 0);
        ScanRequest mansion = request(720, 1146, 100, 3, 120);   // the sparse visits of the MansionLR shape: two waves per pixel
        mansion.ctx_packed = true;
        mansion.ctx_groups = kSweepGroups;
        mansion.precompacted = 2;
        q = plan_scan(mansion, 68);
        CHECK(mansion.use_stream && q.packed && q.px_waves == 2 && q.groups == 1 && q.stream_park == 24);
        // shared taps only where the re-gathered tail is long: 8 of 100 views here (64-pixel tiles), 109 of 201 at c5 (63)
        ScanRequest dense = request(720, 1146, 100, 3, 120);
        CHECK(dense.use_stream && plan_scan(dense, 68).tile_w == 64 && !stream_shares_taps(100, 68, 24));
        dense.stream_share = 2;
        CHECK(plan_scan(dense, 68).tile_w == 63);
        ScanRequest c5s = request(2160, 4096, 201, 3, 512);
        c5s.use_chip = false;
        c5s.use_stream = true;
        CHECK(plan_scan(c5s, 68).tile_w == 63 && stream_shares_taps(201, 68, 24));
        c5s.stream_share = 0;
        CHECK(plan_scan(c5s, 68).tile_w == 64);
    }
    // K1 left row lists: never packed, whatever the caller asked for
    sparse.precompacted = 1;
    p = plan_scan(sparse, 0);
    CHECK(!p.packed);
    // more pixels than an int counts: never packed
    ScanRequest huge = request(60000, 40000, 33, 1, 128);
    huge.ctx_packed = true;
    CHECK(!plan_scan(huge, 0).packed);

    std::mt19937 rng(7);
    for (int it = 0; it < 50000; it++) {
        const int C = (rng() & 1) ? 1 : 3;
        ScanRequest r = request(1 + (int)(rng() % 2500), 1 + (int)(rng() % 5000), 1 + (int)(rng() % 300), C, 2 + (int)(rng() % 600));
        r.reg_waves = r.spad ? 1 + (int)(rng() % 8) : 0;
        r.num_cus = (rng() % 8 == 0) ? 0 : 256;
        r.ctx_groups = (rng() % 4 == 0) ? kSweepGroups : 1;
        r.ctx_packed = r.ctx_groups > 1;
        r.precompacted = (int)(rng() % 3);
        r.force_groups = (rng() % 5 == 0) ? (int)(rng() % 70) : 0;
        r.force_packed = (int)(rng() % 3) - 1;
        r.stream_groups = (rng() % 6 == 0) ? (int)(rng() % 40) : 0;
        r.stream_share = (int)(rng() % 3);
        r.stream_lds_bytes = (size_t)(16 + rng() % 137) << 10;
        if (rng() % 7 == 0) {   // forced streaming kernel on a shape the register kernel would take
            r.spad = 0;
            r.use_stream = true;
            r.reg_waves = 0;
        }
        const int nres = r.use_stream ? (C == 1 ? (r.S >= 192 ? 192 : 0) : (r.S >= 68 ? 68 : r.S >= 48 ? 48 : 0)) : 0;
        check_plan(r, plan_scan(r, nres));
        size_t recs, tickets;
        sweep_record_plan((size_t)r.V * r.U, r.dim_d, r.use_stream, &recs, &tickets);
        if (recs)
            CHECK(tickets > 0 && (r.use_stream ? recs * kPartialRecordBytes <= kPartialBudget || true : recs <= (size_t)kPackedItemTarget * 64));
    }
    // what the sweep sizes in advance is what its sparse visits ask for
    for (int D : {8, 16, 64, 128, 256, 512}) {
        ScanRequest r = request(512, 512, 33, 1, D);
        r.ctx_groups = kSweepGroups;
        r.ctx_packed = true;
        r.precompacted = 2;
        const ScanPlan q = plan_scan(r, 0);
        size_t recs, tickets;
        sweep_record_plan((size_t)512 * 512, D, false, &recs, &tickets);
        CHECK(q.groups == 1 ? recs == 0 || recs >= q.records : (recs == q.records && tickets == q.tickets));
    }
    for (int U : {1, 2, 63, 64, 65, 1920, 4096, 1 << 20, (1 << 24) - 3}) {
        const float f = stream_frac_max(U);
        CHECK(f < 1.0f && (f > 0.0f || U >= (1 << 23)));   // rows too long for fractions to exist: no fraction qualifies
        // a fraction at or below f added to the largest position cannot round up to the next integer
        const float x = (float)(U + 1) + f;
        CHECK(x < (float)(U + 2) || (float)(U + 1) + 1.0f == (float)(U + 1));
    }
}

static void test_chip_ladder()
{
    // rungs ascend, at most kChipPadMax views apart, each within the tiers' capacity; the top one is c5's 201 views
    for (int i = 0; i < kChipRungs; i++) {
        const ChipRung r = kChipLadder[i];
        CHECK(r.na % 4 == 0 && r.na <= kChipNAMax && r.nl % 2 == 0 && r.nl <= kChipNLMax && (r.nl == 0 || r.nl >= 8));
        CHECK(r.nl == 0 || r.na == kChipNAMax);   // the LDS tier only behind a full AGPR tier
        if (i > 0)
            CHECK(r.views() > kChipLadder[i - 1].views() && r.views() - kChipLadder[i - 1].views() <= kChipPadMax);
    }
    CHECK(kChipTopS == 201 && kChipLadder[kChipRungs - 1].na == 84 && kChipLadder[kChipRungs - 1].nl == 50);
    // every view count the kernel takes has a rung that holds it with at most kChipPadMax views of padding -- or is beyond
    // the top rung, which then fetches the rest per pass -- and its LDS share fits; no other count is taken
    for (int S = 1; S <= 300; S++) {
        const int k = chip_rung_for(S);
        CHECK((k >= 0) == (S >= kChipMinS && S <= kChipMaxS));
        CHECK(chip_takes(S, 3) == (k >= 0) && !chip_takes(S, 1));
        if (k < 0)
            continue;
        const ChipRung r = kChipLadder[k];
        if (S <= kChipTopS) {
            CHECK(r.views() >= S && r.views() - S <= kChipPadMax);
            CHECK(k == 0 || kChipLadder[k - 1].views() < S);   // the smallest such rung
            CHECK(chip_table_floats(S, r.views()) >= r.views());
        } else {
            CHECK(k == kChipRungs - 1 && chip_table_floats(S, r.views()) >= S);
        }
        CHECK((size_t)chip_wave_floats(S, r) * 4 * kScanWavesPerTile <= kChipLdsBytes);
        CHECK(chip_wave_floats(S, r) >= 2 * (64 + 6 * 32));   // the epilogue's block reuses the region's head
    }
    CHECK(kChipMinS == 123 && chip_rung_for(123) == 0 && kChipLadder[0].views() == 127);
    CHECK(chip_rung_for(201) == kChipRungs - 1 && chip_rung_for(220) == kChipRungs - 1 && chip_rung_for(200) == kChipRungs - 1);
    CHECK(kChipLadder[chip_rung_for(150)].views() == 151 && kChipLadder[chip_rung_for(152)].views() == 159);
}

static void test_visit_schedule()
{
    for (int nd = 1; nd <= 9; nd++)
        for (int h = 0; h <= 3; h++) {
            const std::vector<VisitOp> ops = sweep_visit_schedule(nd, h);
            // replay: an op may run once everything it waits for has been queued BEFORE it (one host thread queues in order)
            std::vector<int> scanned(nd, 0), fetched_from_lo(nd, 0), fetched_from_hi(nd, 0), finished(nd, 0);
            for (const VisitOp& op : ops) {
                CHECK(op.dev >= 0 && op.dev < nd);
                for (int k : op.wait_scan_of)
                    CHECK(scanned[k]);
                if (op.kind == VisitOp::SCAN) {
                    CHECK(!scanned[op.dev]);
                    scanned[op.dev] = 1;
                } else if (op.kind == VisitOp::FETCH) {
                    CHECK(h > 0 && scanned[op.dev] && (op.neighbour == op.dev - 1 || op.neighbour == op.dev + 1));
                    CHECK(scanned[op.neighbour] && !finished[op.neighbour]);   // reads the neighbour's RAW rows
                    CHECK(op.wait_scan_of.size() == 1 && op.wait_scan_of[0] == op.neighbour);
                    (op.neighbour < op.dev ? fetched_from_lo : fetched_from_hi)[op.dev] = 1;
                } else {
                    CHECK(scanned[op.dev] && !finished[op.dev]);
                    // my own halo rows are in place, and both neighbours have taken my rows before I rewrite them
                    if (h > 0 && op.dev > 0)
                        CHECK(fetched_from_lo[op.dev] && fetched_from_hi[op.dev - 1]);
                    if (h > 0 && op.dev + 1 < nd)
                        CHECK(fetched_from_hi[op.dev] && fetched_from_lo[op.dev + 1]);
                    std::set<int> w(op.wait_fetch_of.begin(), op.wait_fetch_of.end());
                    CHECK((int)w.size() == (op.dev > 0) + (op.dev + 1 < nd));
                    finished[op.dev] = 1;
                }
            }
            for (int i = 0; i < nd; i++)
                CHECK(scanned[i] && finished[i]);
        }
    // the rows a fetch moves: always rows the neighbour OWNS, into halo rows the fetcher does NOT own
    for (int V : {16, 135, 1080})
        for (int nd : {2, 3, 8})
            for (int h : {1, 2, 3}) {
                if (sweep_devices_for(V, nd, h) != nd)
                    continue;
                for (int i = 0; i < nd; i++)
                    for (int side = 0; side < 2; side++) {
                        const int k = side == 0 ? i - 1 : i + 1;
                        if (k < 0 || k >= nd)
                            continue;
                        const RowBlock bd = row_block(V, i, nd, h), bo = row_block(V, k, nd, h);
                        int dst, src;
                        fetch_rows(bd, bo, side, h, &dst, &src);
                        for (int r = 0; r < h; r++) {
                            const int gd = bd.lo + dst + r, gs = bo.lo + src + r;   // global scanline numbers
                            CHECK(gd == gs);                                          // same scanline on both sides
                            CHECK(gs >= bo.a && gs < bo.b);                           // the neighbour owns it
                            CHECK(gd >= bd.lo && gd < bd.hi && !(gd >= bd.a && gd < bd.b));   // a halo row of mine
                        }
                    }
            }
}

static void test_median_plan()
{
    // core.hpp:686: width = (size - 1) / 2 -- an even size is the next smaller odd window, 0 the 1 x 1 window
    CHECK(median_width(5) == 2 && median_width(4) == 1 && median_width(6) == 2 && median_width(11) == 5 && median_width(1) == 0);
    CHECK(median_width(0) == 0 && median_width(2) == 0 && median_width(15) == 7 && median_width(16) == 7);
    CHECK(median_halo(11) == 5 && median_halo(0) == 0 && median_halo(6) == 2 && halo_rows(11, 1) == 5 && halo_rows(4, 3) == 3);
    for (int C : {1, 3})
        for (int size = 0; size <= 80; size++) {
            const MedianPlan m = median_plan(size, C);
            CHECK(m.w == std::max(0, (size - 1) / 2));
            const int side = 2 * m.w + 1;
            if (side <= kMedianNetMaxSide)
                CHECK(m.mode == side);
            else if (side <= kMedianTileMaxSide)
                CHECK(m.mode == 0);
            else
                CHECK(m.mode == -1 && m.lds_bytes == 0);
            if (m.mode >= 0) {
                CHECK(m.lds_bytes == (size_t)(1 + C) * side * (256 + 2 * m.w) * 4);
                CHECK(m.lds_bytes <= (size_t)160 << 10);   // a CU's LDS
            }
        }
    CHECK(median_plan(5, 1).mode == 5 && median_plan(11, 3).mode == 11 && median_plan(12, 3).mode == 11 && median_plan(15, 3).mode == 0);
    CHECK(median_plan(32, 3).mode == 0 && median_plan(33, 3).mode == -1);
}

static float norm1_ref(float x) { return (float)((double)std::fabs(x) * 1.73205080757); }                          // rslf_types.cpp:80-84
static float norm3_ref(float x, float y, float z) { double s = (double)x * x; s += (double)y * y; s += (double)z * z; return (float)std::sqrt(s); }

static void test_norm_thresholds()
{
    // norm<T>(x) < eps  <=>  |x| < a1  /  sum of squares < s3, exactly: on both sides of the thresholds and at random
    std::mt19937 rng(20260411);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f);
    const float inf = std::numeric_limits<float>::infinity();
    std::vector<float> epss = {0.1f, 0.25f, 0.05f, 1.0f, 10.0f, 1e-30f, 1e30f, 3.0e38f, inf, 0.0f, -1.0f, std::nanf(""), 1.17549435e-38f, 1e-44f};
    for (int i = 0; i < 40; i++)
        epss.push_back(std::ldexp(u01(rng) + 0.5f, (int)(u01(rng) * 60) - 40));
    for (float eps : epss) {
        const NormThreshold t = norm_threshold(eps);
        CHECK(t.a1 >= 0.0f && t.s3 >= 0.0);
        // neighbours of the thresholds
        for (int d = -3; d <= 3; d++) {
            uint32_t b;
            memcpy(&b, &t.a1, 4);
            if ((long long)b + d >= 0 && (long long)b + d <= 0x7F800000ll) {
                const uint32_t bb = (uint32_t)((long long)b + d);
                float x;
                memcpy(&x, &bb, 4);
                CHECK((norm1_ref(x) < eps) == (std::fabs(x) < t.a1));
                CHECK((norm1_ref(-x) < eps) == (std::fabs(-x) < t.a1));
            }
            uint64_t q;
            memcpy(&q, &t.s3, 8);
            if (d >= 0 || q >= (uint64_t)(-d)) {
                const uint64_t qq = q + (uint64_t)(long long)d;
                if (qq <= 0x7FF0000000000000ull) {
                    double s;
                    memcpy(&s, &qq, 8);
                    CHECK(((float)std::sqrt(s) < eps) == (s < t.s3));
                }
            }
        }
        for (int it = 0; it < 2000; it++) {
            const float scale = std::isfinite(eps) && eps > 0 ? eps : 1.0f;
            const float x = (u01(rng) * 2 - 1) * scale * (it % 3 == 0 ? 0.58f : 1.5f);
            const float y = (u01(rng) * 2 - 1) * scale * 0.7f, z = (u01(rng) * 2 - 1) * scale * 0.7f;
            CHECK((norm1_ref(x) < eps) == (std::fabs(x) < t.a1));
            double s = (double)x * x;
            s += (double)y * y;
            s += (double)z * z;
            CHECK((norm3_ref(x, y, z) < eps) == (s < t.s3));
        }
        const float nan = std::nanf("");
        CHECK(!(std::fabs(nan) < t.a1) && !((double)nan < t.s3));   // NaN fails, as in the reference
        CHECK(!(std::fabs(inf) < t.a1));
    }
    // the defaults: median_filter_epsilon 0.1, propagation_epsilon 0.1 -> |x| < 0.1 / 1.7320508...
    const NormThreshold d = norm_threshold(0.1f);
    CHECK(std::fabs(d.a1 - 0.0577350f) < 1e-6f && std::fabs(d.s3 - 0.01) < 1e-8);
}

int main()
{
    test_median_plan();
    test_norm_thresholds();
    test_sweep_order();
    test_small_rules();
    test_pyramid();
    test_structuring_element();
    test_partitions();
    test_host_copies();
    test_plane_layout();
    test_scan_plans();
    test_chip_ladder();
    test_visit_schedule();
    std::printf("plan tests ok: %d checks\n", g_checks);
    return 0;
}
