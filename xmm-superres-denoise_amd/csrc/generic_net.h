// generic_net.h -- the RRDB generators at arbitrary widths on exact-fp32 direct-convolution HIP kernels (generic_net.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <vector>

#include "../../include/xsd.h"

namespace xsd {

struct GConvW {              // flat-param offsets of weight / bias, offset of the transposed copy, offsets of the block-packed copies
    long long w, b, t, pf, pt;
    int cout, cin;
    bool wide() const { return cout >= 16 && cin >= 16; }     // runs on the fp32 matrix instruction
};

struct GenericNet {
    static constexpr int MAX_PARTS = 64;
    xsd_config cfg;
    int nf = 0, cin = 0, cout = 0, blocks = 0, nup = 0;
    bool sr = false;
    long long nparams = 0, wt_floats = 0;
    GConvW first, trunk, last, hr;
    std::vector<GConvW> rdb, up;
    std::vector<long long> rrdb_begin;
    int ndesc = 0, max_w = 0;
    float* wt = nullptr;             // transposed + flipped weights (input-gradient convs), rebuilt by pack()
    float* wblk = nullptr;           // [co block][ci block][tap][32][32] copies of the wide convs and their input-gradient convs
    long long wblk_floats = 0;
    void* descs_dev = nullptr;
    const float* params = nullptr;   // borrowed
    bool packed = false, saved = false;
    // plan / workspace
    int B = 0, H = 0, W = 0, train = -1;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    float *fea = nullptr, *rin = nullptr, *rout = nullptr, *T = nullptr, *H1 = nullptr, *pre = nullptr;
    std::vector<float*> slabs, U;
    float *dpre = nullptr, *dS[2] = {nullptr, nullptr}, *gtmp = nullptr, *dRR = nullptr, *dT = nullptr, *dHi[2] = {nullptr, nullptr}, *gup = nullptr;
    float *wg_partial = nullptr, *wg_bias_partial = nullptr;
    const float* b_x = nullptr;

    static GenericNet* create(const xsd_config& cfg);
    ~GenericNet();
    hipError_t pack(const float* dev_params, hipStream_t s);
    hipError_t forward(const float* x, float* y, int B, int H, int W, bool save, hipStream_t s);
    int num_stages() const { return blocks + 2; }
    hipError_t backward_stage(int stage, const float* dy, float* dx, float* grads, hipStream_t s);
    void grad_range(int stage, long long* off, long long* cnt) const;

private:
    hipError_t plan(int B, int H, int W, bool train);
    hipError_t conv(hipStream_t s, const GConvW& c, bool transposed, const float* x, long long xbs, float* y, long long ybs, int B, int H, int W,
                    const std::function<void(void*)>& tweak);
    hipError_t wgrad(hipStream_t s, const GConvW& c, const float* x, long long xbs, const float* g, long long gbs, int B, int H, int W, float* grads);
    hipError_t ew(hipStream_t s, int op, float* d, long long dbs, const float* src, long long sbs, int C, long long HW, float a);
    hipError_t rdb_backward(hipStream_t s, int k, const float* dOut, long long dOut_bs, float* dSk, float* grads);
};

} // namespace xsd
