// Exercises include/rslf_hip.hpp the way the reference's demo uses its class
// (RSLightFields/tests/test_depth_computation_pile.cpp:49-51):
//     rslf::Depth1DComputer_pile<float> depth_computer_1d(epis, d_min, d_max, dim_d);
//     depth_computer_1d.run();
// Built with g++ -std=c++11 (the reference's toolchain) against librslf_hip.so; no OpenCV here,
// so the EPIs are plain buffers.  Writes input and results to <out_dir>/ for the pytest side
// (tests/test_gpu_cpp_host.py) to compare against the oracle.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rslf_hip.hpp"

template <typename T>
static void dump(const std::string& path, const std::vector<T>& v)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) {
        std::perror(path.c_str());
        std::exit(2);
    }
    std::fclose(f);
}

template <int C>
static int run_case(rslfx::Context& ctx, const std::string& dir, const std::string& tag, bool u8)
{
    const int V = 7, S = 13, U = 150, D = 20;
    const float dmin = -1.5f, dmax = 2.0f;
    // deterministic pseudo-random EPIs, one separately allocated buffer per scanline (a Vec<Mat>)
    std::vector<std::vector<float> > epis_f(V);
    std::vector<std::vector<unsigned char> > epis_u8(V);
    std::vector<const void*> ptrs(V);
    unsigned state = 12345u + (unsigned)C + (u8 ? 77u : 0u);
    std::vector<float> flat;
    for (int v = 0; v < V; v++) {
        epis_f[v].resize((size_t)S * U * C);
        epis_u8[v].resize((size_t)S * U * C);
        for (size_t i = 0; i < epis_f[v].size(); i++) {
            state = state * 1664525u + 1013904223u;
            const unsigned r = (state >> 8) & 0xffffu;
            epis_u8[v][i] = (unsigned char)(r & 0xffu);
            epis_f[v][i] = 3.0f + 250.0f * (float)r / 65535.0f;   // raw sensor-like range: the ctor rescales by the max
            flat.push_back(u8 ? (float)epis_u8[v][i] : epis_f[v][i]);
        }
        ptrs[v] = u8 ? (const void*)epis_u8[v].data() : (const void*)epis_f[v].data();
    }
    rslfx::Depth1DParameters params;   // defaults = the reference's
    rslfx::Depth1DComputer_pile<C> depth_computer_1d(ctx, ptrs.data(), u8, V, S, U, 0, dmin, dmax, D, -1, -1.0f, params);
    depth_computer_1d.run();
    if (depth_computer_1d.get_s_hat() != S / 2)
        return 1;
    dump(dir + "/" + tag + "_input.f32", flat);
    dump(dir + "/" + tag + "_Ce.f32", depth_computer_1d.m_edge_confidence_v_u);
    dump(dir + "/" + tag + "_mask.u8", depth_computer_1d.m_edge_confidence_mask_v_u);
    dump(dir + "/" + tag + "_Cd.f32", depth_computer_1d.m_disp_confidence_v_u);
    dump(dir + "/" + tag + "_depth.f32", depth_computer_1d.m_best_depth_v_u);
    dump(dir + "/" + tag + "_rbar.f32", depth_computer_1d.m_rbar_v_u);
    dump(dir + "/" + tag + "_idx.i32", depth_computer_1d.m_depth_idx_v_u);
    dump(dir + "/" + tag + "_score.f32", depth_computer_1d.m_score_v_u);
    std::printf("%s V=%d S=%d U=%d C=%d D=%d scale=%.9g scanned=%lld kernel=%d spad=%d K2=%.3f ms\n", tag.c_str(), V, S, U, C, D,
                depth_computer_1d.epi_scale_factor(), (long long)depth_computer_1d.stats.pixels_scanned,
                depth_computer_1d.stats.scan_kernel, depth_computer_1d.stats.s_pad, ctx.last_scan_kernel_ms());
    return 0;
}

// The rows around the path through the same header: Depth2DComputer and FineToCoarse on plain buffers.
static int run_sweeps(rslfx::Context& ctx, const std::string& dir)
{
    const int V = 44, S = 5, U = 64, D = 9;
    std::vector<std::vector<float> > epis(V);
    std::vector<const void*> ptrs(V);
    std::vector<float> flat;
    unsigned state = 777u;
    for (int v = 0; v < V; v++) {
        epis[v].resize((size_t)S * U);
        // a little structure: the same texture in every view, shifted by one column per view on the lower half
        std::vector<float> tex(U + 2 * S);
        for (size_t i = 0; i < tex.size(); i++) {
            state = state * 1664525u + 1013904223u;
            tex[i] = 3.0f + 200.0f * (float)((state >> 8) & 0xffffu) / 65535.0f;
        }
        for (int s = 0; s < S; s++)
            for (int u = 0; u < U; u++)
                epis[v][(size_t)s * U + u] = tex[u + S + ((v >= V / 2) ? (s - S / 2) : 0)];
        ptrs[v] = epis[v].data();
        flat.insert(flat.end(), epis[v].begin(), epis[v].end());
    }
    dump(dir + "/sweep_input.f32", flat);
    rslfx::Depth2DComputer<1> d2(ctx, ptrs.data(), false, V, S, U, 0, -1.0f, 1.0f, D);
    d2.run();
    dump(dir + "/d2_depth.f32", d2.get_depths_s_v_u());
    dump(dir + "/d2_mask.u8", d2.m_edge_confidence_mask_s_v_u);
    dump(dir + "/d2_Ce.f32", d2.m_edge_confidence_s_v_u);
    rslfx::FineToCoarse<1> f2c(ctx, ptrs.data(), false, V, S, U, 0, -1.0f, 1.0f, D);
    f2c.run();
    std::vector<float> map;
    std::vector<uint8_t> valid;
    f2c.get_results(map, valid);
    dump(dir + "/f2c_map.f32", map);
    dump(dir + "/f2c_valid.u8", valid);
    std::printf("sweeps: Depth2DComputer scanned %lld px, FineToCoarse %d levels, %lld px\n", (long long)d2.stats.pixels_scanned,
                f2c.pyramid_depth(), (long long)f2c.stats.pixels_scanned);
    // the sweep over a MultiContext (three workers on the one GPU, a neighbour exchange per visit): identical planes
    rslfx::MultiContext multi(std::vector<int>(3, 0));
    rslfx::Depth2DComputer<1> d2m(multi, ptrs.data(), false, V, S, U, 0, -1.0f, 1.0f, D);
    d2m.run();
    const bool same = d2m.get_depths_s_v_u() == d2.get_depths_s_v_u() && d2m.m_edge_confidence_mask_s_v_u == d2.m_edge_confidence_mask_s_v_u &&
                      d2m.m_edge_confidence_s_v_u == d2.m_edge_confidence_s_v_u && d2m.m_disp_confidence_s_v_u == d2.m_disp_confidence_s_v_u &&
                      d2m.m_rbar_s_v_u == d2.m_rbar_s_v_u && d2m.stats.pixels_scanned == d2.stats.pixels_scanned;
    std::printf("sweeps: Depth2DComputer over %d workers: %s\n", multi.device_count(), same ? "planes identical" : "MISMATCH");
    if (!same)
        return 32;
    rslfx::FineToCoarse<1> f2cm(multi, ptrs.data(), false, V, S, U, 0, -1.0f, 1.0f, D);
    f2cm.run();
    std::vector<float> map_m;
    std::vector<uint8_t> valid_m;
    f2cm.get_results(map_m, valid_m);
    const bool same_f2c = map_m == map && valid_m == valid && f2cm.pyramid_depth() == f2c.pyramid_depth() &&
                          f2cm.stats.pixels_scanned == f2c.stats.pixels_scanned;
    std::printf("sweeps: FineToCoarse over %d workers: %s\n", multi.device_count(), same_f2c ? "fused map identical" : "MISMATCH");
    if (!same_f2c)
        return 64;
    return f2c.pyramid_depth() == 3 ? 0 : 8;   // 44x64 -> 22x32 -> 11x16, then 6x8 stops the pyramid
}

// The multi-device form (rslfx::MultiContext): two workers on ONE GPU, three scanlines per chunk -- every plane must
// equal the single-context run bit for bit, default normalisation (the maximum over ALL EPIs) included.
template <int C>
static int run_multi(rslfx::Context& ctx, bool u8)
{
    const int V = 29, S = 9, U = 96, D = 14;
    std::vector<std::vector<float> > epis_f(V);
    std::vector<std::vector<unsigned char> > epis_u8(V);
    std::vector<const void*> ptrs(V);
    unsigned state = 4242u + (unsigned)C;
    for (int v = 0; v < V; v++) {
        epis_f[v].resize((size_t)S * U * C);
        epis_u8[v].resize((size_t)S * U * C);
        for (size_t i = 0; i < epis_f[v].size(); i++) {
            state = state * 1664525u + 1013904223u;
            const unsigned r = (state >> 8) & 0xffffu;
            epis_u8[v][i] = (unsigned char)(r & 0xffu);
            // the brightest rows are the last ones: a block-local maximum would rescale the first block
            epis_f[v][i] = (v < V / 2 ? 40.0f : 3.0f) + (v < V / 2 ? 60.0f : 250.0f) * (float)r / 65535.0f;
        }
        ptrs[v] = u8 ? (const void*)epis_u8[v].data() : (const void*)epis_f[v].data();
    }
    rslfx::Depth1DComputer_pile<C> one(ctx, ptrs.data(), u8, V, S, U, 0, -1.0f, 2.0f, D);
    one.run();
    rslfx::MultiContext multi(std::vector<int>(2, 0));
    multi.set_chunk_rows(3);
    rslfx::Depth1DComputer_pile<C> two(multi, ptrs.data(), u8, V, S, U, 0, -1.0f, 2.0f, D);
    two.run();
    int bad = 0;   // one bit per plane / figure that differs
    bad |= (one.m_edge_confidence_v_u != two.m_edge_confidence_v_u) << 0;
    bad |= (one.m_edge_confidence_mask_v_u != two.m_edge_confidence_mask_v_u) << 1;
    bad |= (one.m_disp_confidence_v_u != two.m_disp_confidence_v_u) << 2;
    bad |= (one.m_best_depth_v_u != two.m_best_depth_v_u) << 3;
    bad |= (one.m_rbar_v_u != two.m_rbar_v_u) << 4;
    bad |= (one.m_depth_idx_v_u != two.m_depth_idx_v_u) << 5;
    bad |= (one.m_score_v_u != two.m_score_v_u) << 6;
    bad |= (one.stats.pixels_scanned != two.stats.pixels_scanned) << 7;
    bad |= (one.epi_scale_factor() != two.epi_scale_factor()) << 8;
    std::printf("multi C=%d %s: %d devices, %lld px scanned (single context %lld), scale %.9g (%.9g), %s 0x%x\n", C,
                u8 ? "u8" : "f32", multi.device_count(), (long long)two.stats.pixels_scanned, (long long)one.stats.pixels_scanned,
                two.epi_scale_factor(), one.epi_scale_factor(), bad ? "MISMATCH" : "planes identical", bad);
    return bad ? 16 : 0;
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    try {
        rslfx::Context ctx(0);
        int rc = 0;
        rc |= run_case<1>(ctx, dir, "f32_1ch", false);
        rc |= run_case<3>(ctx, dir, "f32_3ch", false);
        rc |= run_case<3>(ctx, dir, "u8_3ch", true);
        rc |= run_sweeps(ctx, dir);
        rc |= run_multi<1>(ctx, false);
        rc |= run_multi<3>(ctx, true);
        // error convention: the C-ABI never throws; the C++ wrapper turns its status into rslfx::Error
        bool threw = false;
        try {
            std::vector<float> e(4 * 8, 0.5f);
            const void* p[1] = {e.data()};
            rslfx::Depth1DComputer_pile<1> bad(ctx, p, false, 1, 4, 8, 0, 0.f, 1.f, 1 /* dim_d < 2 */);
            bad.run();
        } catch (const rslfx::Error& err) {
            threw = err.status == RSLF_ERR_INVALID_ARG;
        }
        if (!threw) {
            std::fprintf(stderr, "expected RSLF_ERR_INVALID_ARG for dim_d = 1\n");
            rc |= 4;
        }
        return rc;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "FAILED: %s\n", e.what());
        return 3;
    }
}
