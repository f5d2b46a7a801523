// probe (round 3): where does the fused NeRF head forward spend its time?  Compiles csrc/ffmlp.hip with in-kernel stamps
// (100 MHz wall clock + shader clock per wave: start, every 64-row group, end) and prints the distribution.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DLAE_MLP_STAMPS \
//         -I include tools/ubench/mlp_probe.hip laenerf_amd/csrc/lae_common.cpp -o tools/ubench/bin/mlp_probe
#include "../../laenerf_amd/csrc/ffmlp.hip"
extern "C" int lae_loss_finish(const float*, uint32_t, uint32_t, const float*, float*, void*) { return 0; }   // lives in raymarching.hip; not used here
#include <stdio.h>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? (uint32_t)atoi(argv[1]) : 257792u;
    const int bpc = argc > 2 ? atoi(argv[2]) : 1;
    const int variant = argc > 3 ? atoi(argv[3]) : 4;          // 4: k_nerf_head_fwd4; 54 / 58: k_nerf_head_fwd5 with 4 / 8 waves per workgroup
    std::vector<uint16_t> h_enc((size_t)M * 32), h_ws(64 * 112), h_wc(64 * 176);
    std::vector<float> h_dirs((size_t)M * 3);
    uint32_t x = 1;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (float)(x >> 8) / 16777216.0f * 2.0f - 1.0f; };
    auto f2h = [](float f) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); };
    for (auto& v : h_enc) v = f2h(rnd() * 0.5f);
    for (auto& v : h_ws) v = f2h(rnd() * 0.2165f);
    for (auto& v : h_wc) v = f2h(rnd() * 0.2165f);
    for (size_t i = 0; i < M; i++) { float a = rnd(), b = rnd(), c = rnd(); float n = sqrtf(a * a + b * b + c * c) + 1e-9f; h_dirs[3 * i] = a / n; h_dirs[3 * i + 1] = b / n; h_dirs[3 * i + 2] = c / n; }
    half_t *enc, *ws, *wc, *hout; float *dirs, *sig, *rgb;
    hipMalloc(&enc, h_enc.size() * 2); hipMalloc(&ws, h_ws.size() * 2); hipMalloc(&wc, h_wc.size() * 2); hipMalloc(&hout, (size_t)M * 32);
    hipMalloc(&dirs, h_dirs.size() * 4); hipMalloc(&sig, (size_t)M * 4); hipMalloc(&rgb, (size_t)M * 12);
    hipMemcpy(enc, h_enc.data(), h_enc.size() * 2, hipMemcpyHostToDevice); hipMemcpy(ws, h_ws.data(), h_ws.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(wc, h_wc.data(), h_wc.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dirs, h_dirs.data(), h_dirs.size() * 4, hipMemcpyHostToDevice);
    const uint32_t n_tiles = M / 16, n_groups = (n_tiles + 3) / 4;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const uint32_t WV = variant == 58 ? 8 : 4;
    const uint32_t blocks = std::min((n_groups + WV - 1) / WV, (uint32_t)p.multiProcessorCount * bpc);
    const size_t lds5 = (size_t)Head5Img::END * 2 + WV * sizeof(Head4Scratch);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nerf_head_fwd5<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nerf_head_fwd5<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; k++) {
            if (variant == 54) k_nerf_head_fwd5<true, 4><<<blocks, 256, lds5>>>(enc, dirs, ws, wc, n_tiles, 1.0f, hout, sig, rgb, 1, nullptr, M);
            else if (variant == 58) k_nerf_head_fwd5<true, 8><<<blocks, 512, lds5>>>(enc, dirs, ws, wc, n_tiles, 1.0f, hout, sig, rgb, 1, nullptr, M);
            else k_nerf_head_fwd4<true><<<blocks, 256, 4 * sizeof(Head4Scratch)>>>(enc, dirs, ws, wc, n_tiles, 1.0f, hout, sig, rgb, 1, nullptr, M);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("variant %d: %u blocks x %u waves, %.2f us per launch (10 back to back)\n", variant, blocks, WV, ms * 100.0f);
    }
    if (variant >= 100) {
        // backward probe: k_mlp_bwd_wave<32, 2, 1, 1> (colour net of the fused head; 101) or <32, 1, 0> (sigma net; 100)
        float *gs, *gr; half_t *gh, *genc; float* slabs;
        hipMalloc(&gs, (size_t)M * 4); hipMalloc(&gr, (size_t)M * 12); hipMalloc(&gh, (size_t)M * 32); hipMalloc(&genc, (size_t)M * 64);
        hipMemset(gs, 0, (size_t)M * 4); hipMemset(gr, 0, (size_t)M * 12); hipMemset(gh, 0, (size_t)M * 32);
        k_nerf_head_fwd5<true, 8><<<256, 512, lds5>>>(enc, dirs, ws, wc, n_tiles, 1.0f, hout, sig, rgb, 1, nullptr, M);
        const uint32_t nb = std::min(((n_tiles + 1) / 2 + 3) / 4, (uint32_t)p.multiProcessorCount);
        const uint32_t nWc = 64 * (32 + 128 + 16), nWs = 64 * (32 + 64 + 16);
        hipMalloc(&slabs, (size_t)nb * nWc * 4);
        HeadBwdArgs ha{dirs, rgb, gr, gs, 1.0f, 0}, hs{}; hs.level_major = 1;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_bwd_wave<32, 2, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WaveCfg<32, 2>::LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_bwd_wave<32, 1, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WaveCfg<32, 1>::LDS_BYTES);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; k++) {
                if (variant == 101) k_mlp_bwd_wave<32, 2, 1, 1><<<nb, 256, WaveCfg<32, 2>::LDS_BYTES>>>(nullptr, hout, wc, n_tiles, gh, slabs, nWc, ha);
                else k_mlp_bwd_wave<32, 1, 0, 2><<<nb, 256, WaveCfg<32, 1>::LDS_BYTES>>>(gh, enc, ws, n_tiles, genc, slabs, nWs, hs);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("bwd variant %d: %u blocks, %.2f us per launch\n", variant, nb, ms * 100.0f);
        }
        std::vector<unsigned long long> st((size_t)4096 * 64);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_mlp_stamps), st.size() * 8);
        const uint32_t nw = nb * 4, n_pairs = (n_tiles + 1) / 2;
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        std::vector<double> pro, epi, tot; std::vector<std::vector<double>> pr(12), prc(12);
        for (uint32_t w = 0; w < nw; w++) {
            const unsigned long long* s = &st[(size_t)w * 64];
            const uint32_t mine = (n_pairs - w + nw - 1) / nw;
            pro.push_back((double)(s[2] - s[0]) * 0.01); epi.push_back((double)(s[30] - s[28]) * 0.01); tot.push_back((double)(s[30] - s[0]) * 0.01);
            for (uint32_t k = 0; k < mine && k < 12; k++) {
                const unsigned long long b = (k + 1 < mine) ? s[2 * (k + 2)] : s[28], bc = (k + 1 < mine) ? s[2 * (k + 2) + 1] : s[29];
                pr[k].push_back((double)(b - s[2 * (k + 1)]) * 0.01); prc[k].push_back((double)(bc - s[2 * (k + 1) + 1]));
            }
        }
        printf("prologue (weights -> LDS): median %.2f us; epilogue (dW reduction + slab): median %.2f us; wave total median %.2f max %.2f us\n",
               med(pro), med(epi), med(tot), *std::max_element(tot.begin(), tot.end()));
        for (int k = 0; k < 12; k++) if (!pr[k].empty()) printf("pair %d: %zu waves, median %.2f us = %.0f clocks\n", k, pr[k].size(), med(pr[k]), med(prc[k]));
        const char* ph[4] = {"recompute", "output layer (transposes, dWout, dH)", "hidden layers", "input layer + dX"};
        for (int k = 0; k < 4; k++) {
            std::vector<double> v;
            for (uint32_t w = 0; w < nw; w++) { const unsigned long long* s = &st[(size_t)w * 64]; v.push_back((double)(s[2 * (17 + k) + 1] - s[2 * (16 + k) + 1])); }
            printf("  third pair, %s: median %.0f clocks\n", ph[k], med(v));
        }
        return 0;
    }
    std::vector<unsigned long long> st((size_t)4096 * 64);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_mlp_stamps), st.size() * 8);
    const uint32_t nw = blocks * WV;
    unsigned long long t_min = ~0ull, t_max = 0;
    for (uint32_t w = 0; w < nw; w++) { t_min = std::min(t_min, st[(size_t)w * 64]); t_max = std::max(t_max, st[(size_t)w * 64 + 30]); }
    printf("first wave start -> last wave end: %.2f us (wall clock 100 MHz)\n", (double)(t_max - t_min) * 0.01);
    // per-phase medians over waves
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    std::vector<double> start, wl, tot, cyc_tot;
    std::vector<std::vector<double>> grp(8), grpc(8);
    for (uint32_t w = 0; w < nw; w++) {
        const unsigned long long* s = &st[(size_t)w * 64];
        start.push_back((double)(s[0] - t_min) * 0.01);
        wl.push_back((double)(s[2] - s[0]) * 0.01);
        tot.push_back((double)(s[30] - s[0]) * 0.01);
        cyc_tot.push_back((double)(s[31] - s[1]));
        const uint32_t my_groups = (n_groups - w + nw - 1) / nw;
        for (uint32_t k = 0; k < my_groups && k < 8; k++) {
            const unsigned long long a = s[2 * (k + 1)], b = (k + 1 < my_groups) ? s[2 * (k + 2)] : s[30];
            const unsigned long long ac = s[2 * (k + 1) + 1], bc = (k + 1 < my_groups) ? s[2 * (k + 2) + 1] : s[31];
            grp[k].push_back((double)(b - a) * 0.01); grpc[k].push_back((double)(bc - ac));
        }
    }
    printf("wave start offset: median %.2f us, max %.2f us\n", med(start), *std::max_element(start.begin(), start.end()));
    printf("weights + first request (stamp 0 -> first group): median %.2f us\n", med(wl));
    printf("wave total: median %.2f us, max %.2f us; shader clocks median %.0f -> %.2f GHz\n", med(tot), *std::max_element(tot.begin(), tot.end()),
           med(cyc_tot), med(cyc_tot) / (med(tot) * 1e3));
    for (int k = 0; k < 8; k++) if (!grp[k].empty()) printf("group %d: %zu waves, median %.2f us = %.0f clocks\n", k, grp[k].size(), med(grp[k]), med(grpc[k]));
    return 0;
}
