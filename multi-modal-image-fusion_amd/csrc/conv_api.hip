// C-ABI entry points of the generic ConvLayer kernels: argument validation + dispatch between the
// VALU (fp32 parity) and MFMA (bf16 throughput) implementations.
#include "common.hpp"

namespace mmif {
// conv_valu.hip
int conv_valu(bool dgrad, int dtype, int ks, const TV& tin, const TV& tout, const TV& tmask, const float* w,
              const float* bias, int cin, int cout, int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st);
size_t wgrad_valu_workspace(int cin, int cout, int ks);
int wgrad_valu(int dtype, int ks, const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate,
               float* ws, hipStream_t st);
// conv_mfma.hip
bool conv_mfma_supported(bool dgrad, int ks, int cin, int cout);
bool conv_dgrad_dup_supported(int ks, int cin, const TV& tin, const TV& tout);
int conv_dgrad_dup(const TV& tin, const TV& tout, const void* wpk, int cin, int cout, const TV& tdup, const TV& tmask, int frag, hipStream_t st);
int conv_mfma(bool dgrad, int ks, const TV& tin, const TV& tout, const TV& tmask, const void* w_packed, const float* bias,
              int cin, int cout, int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st, bool fold = false,
              bool* folded = nullptr, const TV* told = nullptr);
bool conv_dgrad_onto_supported(int ks, int cin, int cout, const TV& tin, const TV& tout);
bool wgrad_mfma_supported(int ks, int cin, int cout);
size_t wgrad_mfma_workspace(int cin, int cout, int ks);
int wgrad_mfma(int ks, const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws,
               hipStream_t st);
bool bwd_pair_supported(int ks, int cin, int cout);
size_t bwd_pair_workspace(int cin, int cout);
int bwd_pair(const TV& tx, const TV& tg, const TV& tgx, const void* wpk_dgrad, float* dw, float* db, int cin, int cout, int accumulate, float* ws,
             hipStream_t st);
bool bwd_wide_supported(int ks, int cin, int cout);
size_t bwd_wide_signs_bytes(int n, int cin, int h, int w);
int bwd_wide(const TV& tx, const TV& tg, const TV& tgx, const void* wpk_dgrad, float* dw, float* db, int cin, int cout, uint64_t mask_bits,
             int accumulate, float* ws, unsigned char* signs, hipStream_t st, int phase);
// conv_x3.hip
bool conv_x3_supported(bool dgrad, int ks, int cin, int cout, const TV& tin, const TV& tout);
int conv_x3(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int cin, int cout, int relu,
            uint64_t mask_bits, uint64_t accum_bits, hipStream_t st, int ks, const unsigned* signs = nullptr, bool* folded = nullptr);
bool wgrad_x3_supported(int ks, int cin, int cout, const TV& tx, const TV& tg);
size_t wgrad_x3_workspace(int cin, int cout, int ks);
int wgrad_x3(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws, hipStream_t st, int ks, unsigned* signs = nullptr);
size_t x3_signs_bytes(int n, int cb, int h, int w);
}  // namespace mmif

using namespace mmif;

// AUTO: bf16 tensors -> the bf16 MFMA kernels, fp32 tensors -> the split-bf16 MFMA kernels (x3_ok: shape covered and, for forward /
// dgrad, the x3 operand image given), else the fp32 FMA kernels.
static int pick_impl(int impl, int dtype, bool mfma_ok, const char* what, bool x3_ok = false) {
    if (impl == MMIF_IMPL_AUTO) {
        if (dtype == MMIF_BF16) return mfma_ok ? MMIF_IMPL_MFMA : MMIF_IMPL_VALU;
        return x3_ok ? MMIF_IMPL_X3 : MMIF_IMPL_VALU;
    }
    if (impl == MMIF_IMPL_MFMA && (dtype != MMIF_BF16 || !mfma_ok)) {
        set_error("%s: MFMA implementation unavailable for this dtype/shape", what);
        return -1;
    }
    if (impl == MMIF_IMPL_X3) {
        if (dtype != MMIF_F32 || !x3_ok) {
            set_error("%s: split-bf16 (X3) implementation needs fp32 tensors, a 3x3 layer, its x3 operand image and folded gradients", what);
            return -1;
        }
        return impl;
    }
    if (impl != MMIF_IMPL_VALU && impl != MMIF_IMPL_MFMA) {
        set_error("%s: bad impl %d", what, impl);
        return -1;
    }
    return impl;
}

extern "C" int mmif_conv2d_reflect_fwd(const mmif_tensor* x, const float* w, const void* w_packed, const float* bias,
                                       const mmif_tensor* y, int32_t cin, int32_t cout, int32_t ksize, int32_t relu,
                                       int32_t impl, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(y, "y")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "conv2d_reflect_fwd: ksize must be 1 or 3 (got %d)", ksize);
    MMIF_REQUIRE(x->halo == 0 && y->halo == 0, "conv2d_reflect_fwd: activations must have halo 0");
    MMIF_REQUIRE(x->dtype == y->dtype && x->n == y->n && x->h == y->h && x->w == y->w, "conv2d_reflect_fwd: x/y mismatch");
    MMIF_REQUIRE(cin > 0 && (cin + 7) / 8 == x->cb, "conv2d_reflect_fwd: cin=%d does not match x.cb=%d", cin, x->cb);
    MMIF_REQUIRE(cout > 0 && (cout + 7) / 8 == y->cb, "conv2d_reflect_fwd: cout=%d does not match y.cb=%d", cout, y->cb);
    MMIF_REQUIRE(ksize == 1 || (x->h >= 2 && x->w >= 2), "reflect padding needs h,w >= 2");
    TV tx = make_tv(x), ty = make_tv(y);
    const int im = pick_impl(impl, x->dtype, conv_mfma_supported(false, ksize, cin, cout) && w_packed != nullptr, "conv2d_reflect_fwd",
                             x->dtype == MMIF_F32 && w_packed != nullptr && conv_x3_supported(false, ksize, cin, cout, tx, ty));
    if (im < 0) return MMIF_EINVAL;
    if (im == MMIF_IMPL_X3) return conv_x3(false, tx, ty, ty, w_packed, bias, cin, cout, relu, 0, 0, (hipStream_t)stream, ksize);
    if (im == MMIF_IMPL_MFMA) return conv_mfma(false, ksize, tx, ty, ty, w_packed, bias, cin, cout, relu, 0, 0, (hipStream_t)stream);
    MMIF_REQUIRE(w != nullptr, "conv2d_reflect_fwd: VALU path needs the fp32 master weights");
    return conv_valu(false, x->dtype, ksize, tx, ty, ty, w, bias, cin, cout, relu, 0, 0, (hipStream_t)stream);
}

static int dgrad_impl(const char* what, const mmif_tensor* gy, const float* w, const void* w_packed_t, const mmif_tensor* x,
                      const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits, uint64_t accum_bits,
                      int32_t impl, void* stream, bool fold, const mmif_tensor* gx_old = nullptr) {
    if (int rc = validate_tensor(gy, "gy")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize must be 1 or 3 (got %d)", what, ksize);
    MMIF_REQUIRE(gx->halo >= ksize / 2, "%s: gx needs halo >= ksize/2", what);
    MMIF_REQUIRE(gy->dtype == gx->dtype && gy->n == gx->n && gy->h == gx->h && gy->w == gx->w, "%s: gy/gx mismatch", what);
    MMIF_REQUIRE(cin > 0 && (cin + 7) / 8 == gx->cb, "%s: cin=%d does not match gx.cb=%d", what, cin, gx->cb);
    MMIF_REQUIRE(cout > 0 && (cout + 7) / 8 == gy->cb, "%s: cout=%d does not match gy.cb=%d", what, cout, gy->cb);
    TV tg = make_tv(gy), tgx = make_tv(gx), tm = tgx;
    if (mask_bits) {
        MMIF_REQUIRE(x != nullptr, "%s: mask_bits set but x is NULL", what);
        if (int rc = validate_tensor(x, "x")) return rc;
        MMIF_REQUIRE(x->halo == 0 && x->dtype == gx->dtype && x->n == gx->n && x->h == gx->h && x->w == gx->w && x->cb == gx->cb,
                     "%s: x does not match gx", what);
        tm = make_tv(x);
    }
    const int im = pick_impl(impl, gy->dtype, conv_mfma_supported(true, ksize, cin, cout) && w_packed_t != nullptr, what,
                             gy->dtype == MMIF_F32 && w_packed_t != nullptr && conv_x3_supported(true, ksize, cin, cout, tg, tgx));
    if (im < 0) return MMIF_EINVAL;
    bool folded = false;
    int rc;
    TV told;
    if (gx_old != nullptr) {
        if (int rc2 = validate_tensor(gx_old, "gx_old")) return rc2;
        MMIF_REQUIRE(gx_old->dtype == gx->dtype && gx_old->n == gx->n && gx_old->h == gx->h && gx_old->w == gx->w && gx_old->halo == gx->halo &&
                         gx_old->cb == gx->cb,
                     "%s: gx_old must have gx's shape, halo and channel blocks", what);
        MMIF_REQUIRE(im == MMIF_IMPL_MFMA && fold, "%s: gx_old needs the folded MFMA path", what);
        told = make_tv(gx_old);
    }
    if (im == MMIF_IMPL_X3) {
        rc = conv_x3(true, tg, tgx, tm, w_packed_t, nullptr, cin, cout, 0, mask_bits, accum_bits, (hipStream_t)stream, ksize, nullptr, fold ? &folded : nullptr);
    } else if (im == MMIF_IMPL_MFMA) {
        rc = conv_mfma(true, ksize, tg, tgx, tm, w_packed_t, nullptr, cin, cout, 0, mask_bits, accum_bits, (hipStream_t)stream, fold, &folded,
                       gx_old != nullptr ? &told : nullptr);
    } else {
        MMIF_REQUIRE(w != nullptr, "%s: VALU path needs the fp32 master weights", what);
        rc = conv_valu(true, gy->dtype, ksize, tg, tgx, tm, w, nullptr, cin, cout, 0, mask_bits, accum_bits, (hipStream_t)stream);
    }
    if (rc != MMIF_OK || !fold || folded || gx->halo == 0 || ksize == 1) return rc;   // (a 1x1 dgrad never writes the halo)
    return mmif_fold_halo(gx, stream);   // the kernel chosen wrote the padded domain: fold it with the stand-alone kernel
}

extern "C" int mmif_conv2d_reflect_dgrad(const mmif_tensor* gy, const float* w, const void* w_packed_t, const mmif_tensor* x,
                                         const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits,
                                         uint64_t accum_bits, int32_t impl, void* stream) {
    return dgrad_impl("conv2d_reflect_dgrad", gy, w, w_packed_t, x, gx, cin, cout, ksize, mask_bits, accum_bits, impl, stream, false);
}

extern "C" int mmif_conv2d_reflect_dgrad_folded(const mmif_tensor* gy, const float* w, const void* w_packed_t, const mmif_tensor* x,
                                                const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits,
                                                uint64_t accum_bits, int32_t impl, void* stream) {
    return dgrad_impl("conv2d_reflect_dgrad_folded", gy, w, w_packed_t, x, gx, cin, cout, ksize, mask_bits, accum_bits, impl, stream, true);
}

// gx = fold(dgrad(gy)) + gx_old on the channel blocks in accum_bits (plain on the others), masked by mask_bits: the accumulate operand
// comes from ANOTHER tensor (DenseFuse / VIFNet: both encoder branches start from the one gradient of f1 + f2 -- no per-branch copy).
extern "C" int mmif_conv2d_dgrad_onto_supported(const mmif_tensor* gy, const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize) {
    if (validate_tensor(gy, "gy") != MMIF_OK || validate_tensor(gx, "gx") != MMIF_OK) return 0;
    if (gy->dtype != MMIF_BF16 || gx->dtype != MMIF_BF16 || gy->halo != 1 || !(gy->flags & MMIF_T_FOLDED) || gx->halo != 1) return 0;
    return conv_dgrad_onto_supported(ksize, cin, cout, make_tv(gy), make_tv(gx)) ? 1 : 0;
}
extern "C" int mmif_conv2d_reflect_dgrad_folded_onto(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* x, const mmif_tensor* gx_old,
                                                     const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits,
                                                     uint64_t accum_bits, void* stream) {
    MMIF_REQUIRE(gx_old != nullptr && w_packed_t != nullptr, "conv2d_reflect_dgrad_folded_onto: NULL gx_old / operand image");
    return dgrad_impl("conv2d_reflect_dgrad_folded_onto", gy, nullptr, w_packed_t, x, gx, cin, cout, ksize, mask_bits, accum_bits, MMIF_IMPL_MFMA,
                      stream, true, gx_old);
}

// dgrad (folded; its own output neither masked nor accumulated) that also leaves the two masked copies DenseFuse's encoder branches need of
// 16 of its channels (struct DupOut, csrc/conv_mfma.hip) -- replaces mmif_fuse_elem_bwd on those blocks
extern "C" int mmif_conv2d_dgrad_dup_supported(const mmif_tensor* gy, const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize) {
    (void)cout;
    if (validate_tensor(gy, "gy") != MMIF_OK || validate_tensor(gx, "gx") != MMIF_OK) return 0;
    if (gy->dtype != MMIF_BF16 || gx->dtype != MMIF_BF16 || gy->halo != 1 || !(gy->flags & MMIF_T_FOLDED) || gx->halo != 1) return 0;
    return conv_dgrad_dup_supported(ksize, cin, make_tv(gy), make_tv(gx)) ? 1 : 0;
}
extern "C" int mmif_conv2d_reflect_dgrad_folded_dup(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* gx, int32_t cin, int32_t cout,
                                                    int32_t ksize, const mmif_tensor* dup_out, const mmif_tensor* dup_mask, int32_t frag, void* stream) {
    if (int rc = validate_tensor(gy, "gy")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    if (int rc = validate_tensor(dup_out, "dup_out")) return rc;
    if (int rc = validate_tensor(dup_mask, "dup_mask")) return rc;
    MMIF_REQUIRE(w_packed_t != nullptr, "conv2d_reflect_dgrad_folded_dup: NULL operand image");
    MMIF_REQUIRE(gy->n == gx->n && gy->h == gx->h && gy->w == gx->w, "conv2d_reflect_dgrad_folded_dup: gy / gx mismatch");
    MMIF_REQUIRE(cin > 0 && (cin + 7) / 8 == gx->cb && cout > 0 && (cout + 7) / 8 == gy->cb, "conv2d_reflect_dgrad_folded_dup: channel blocks do not match");
    MMIF_REQUIRE(mmif_conv2d_dgrad_dup_supported(gy, gx, cin, cout, ksize), "conv2d_reflect_dgrad_folded_dup: layer / tensors not taken by the DMA-staged dgrad");
    MMIF_REQUIRE(frag >= 0 && 2 * frag + 1 < gx->cb, "conv2d_reflect_dgrad_folded_dup: fragment %d outside the gradient's %d channel blocks", frag, gx->cb);
    MMIF_REQUIRE(dup_out->dtype == MMIF_BF16 && dup_out->halo == 1 && dup_out->n == gx->n && dup_out->h == gx->h && dup_out->w == gx->w &&
                     dup_out->cb >= 2 * frag + 10,
                 "conv2d_reflect_dgrad_folded_dup: dup_out must be a bf16 halo-1 tensor of gx's shape with >= %d channel blocks", 2 * frag + 10);
    MMIF_REQUIRE(dup_mask->dtype == MMIF_BF16 && dup_mask->halo == 0 && dup_mask->n == gx->n && dup_mask->h == gx->h && dup_mask->w == gx->w &&
                     dup_mask->cb >= 2 * frag + 10,
                 "conv2d_reflect_dgrad_folded_dup: dup_mask must be the bf16 halo-0 activations of gx's shape with >= %d channel blocks", 2 * frag + 10);
    return conv_dgrad_dup(make_tv(gy), make_tv(gx), w_packed_t, cin, cout, make_tv(dup_out), make_tv(dup_mask), frag, (hipStream_t)stream);
}

extern "C" size_t mmif_conv2d_wgrad_workspace(int32_t cin, int32_t cout, int32_t ksize) {
    size_t a = wgrad_valu_workspace(cin, cout, ksize);
    const size_t b = wgrad_mfma_workspace(cin, cout, ksize), c = wgrad_x3_workspace(cin, cout, ksize);
    if (b > a) a = b;
    return c > a ? c : a;
}

extern "C" int mmif_conv2d_reflect_wgrad(const mmif_tensor* x, const mmif_tensor* gy, float* dw, float* db, int32_t cin,
                                         int32_t cout, int32_t ksize, int32_t accumulate, void* workspace,
                                         size_t workspace_bytes, int32_t impl, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(gy, "gy")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "conv2d_reflect_wgrad: ksize must be 1 or 3 (got %d)", ksize);
    MMIF_REQUIRE(x->halo == 0, "conv2d_reflect_wgrad: x must be an activation (halo 0)");
    MMIF_REQUIRE(x->dtype == gy->dtype && x->n == gy->n && x->h == gy->h && x->w == gy->w, "conv2d_reflect_wgrad: x/gy mismatch");
    MMIF_REQUIRE(cin > 0 && (cin + 7) / 8 == x->cb, "conv2d_reflect_wgrad: cin=%d does not match x.cb=%d", cin, x->cb);
    MMIF_REQUIRE(cout > 0 && (cout + 7) / 8 == gy->cb, "conv2d_reflect_wgrad: cout=%d does not match gy.cb=%d", cout, gy->cb);
    MMIF_REQUIRE(dw != nullptr, "conv2d_reflect_wgrad: dw is NULL");
    if (workspace_bytes < mmif_conv2d_wgrad_workspace(cin, cout, ksize)) {
        set_error("conv2d_reflect_wgrad: workspace too small (%zu < %zu)", workspace_bytes, mmif_conv2d_wgrad_workspace(cin, cout, ksize));
        return MMIF_EWORKSPACE;
    }
    TV tx = make_tv(x), tg = make_tv(gy);
    const int im = pick_impl(impl, x->dtype, wgrad_mfma_supported(ksize, cin, cout), "conv2d_reflect_wgrad",
                             x->dtype == MMIF_F32 && wgrad_x3_supported(ksize, cin, cout, tx, tg));
    if (im < 0) return MMIF_EINVAL;
    if (im == MMIF_IMPL_X3) return wgrad_x3(tx, tg, dw, db, cin, cout, accumulate, (float*)workspace, (hipStream_t)stream, ksize);
    if (im == MMIF_IMPL_MFMA) return wgrad_mfma(ksize, tx, tg, dw, db, cin, cout, accumulate, (float*)workspace, (hipStream_t)stream);
    return wgrad_valu(x->dtype, ksize, tx, tg, dw, db, cin, cout, accumulate, (float*)workspace, (hipStream_t)stream);
}

// Backward of one thin 3x3 ConvLayer in ONE launch: gx = [x > 0] * dgrad(gy) (folded convention, as mmif_conv2d_reflect_dgrad_folded with
// every channel block masked and none accumulated) AND dw, db (as mmif_conv2d_reflect_wgrad) from a single staging of the g and x tiles.
extern "C" int mmif_conv2d_bwd_pair_supported(int32_t cin, int32_t cout, int32_t ksize) { return bwd_pair_supported(ksize, cin, cout) ? 1 : 0; }

extern "C" int mmif_conv2d_reflect_bwd_pair(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* x, const mmif_tensor* gx, float* dw,
                                            float* db, int32_t cin, int32_t cout, int32_t ksize, int32_t accumulate, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    if (int rc = validate_tensor(gy, "gy")) return rc;
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    MMIF_REQUIRE(bwd_pair_supported(ksize, cin, cout), "conv2d_reflect_bwd_pair: unsupported layer %d -> %d k%d", cin, cout, ksize);
    MMIF_REQUIRE(w_packed_t != nullptr && dw != nullptr, "conv2d_reflect_bwd_pair: NULL operand image / dw");
    MMIF_REQUIRE(gy->dtype == MMIF_BF16 && x->dtype == MMIF_BF16 && gx->dtype == MMIF_BF16, "conv2d_reflect_bwd_pair: bf16 tensors expected");
    MMIF_REQUIRE(gy->halo == 1 && (gy->flags & MMIF_T_FOLDED), "conv2d_reflect_bwd_pair: gy must be a folded halo-1 gradient");
    MMIF_REQUIRE(x->halo == 0 && gx->halo == 1, "conv2d_reflect_bwd_pair: x halo 0, gx halo 1 expected");
    MMIF_REQUIRE(gy->n == x->n && gy->h == x->h && gy->w == x->w && gx->n == x->n && gx->h == x->h && gx->w == x->w, "conv2d_reflect_bwd_pair: shape mismatch");
    MMIF_REQUIRE((cin + 7) / 8 == x->cb && x->cb == gx->cb && (cout + 7) / 8 == gy->cb, "conv2d_reflect_bwd_pair: channel blocks do not match");
    MMIF_REQUIRE(x->h >= 4 && x->w >= 4, "conv2d_reflect_bwd_pair: needs h, w >= 4 (fold steps inside the border tiles)");
    MMIF_REQUIRE((long long)x->cb_total * x->h * x->w < (1ll << 31) && (long long)gy->cb_total * (gy->h + 2) * (gy->w + 2) < (1ll << 31),
                 "conv2d_reflect_bwd_pair: one image of x / gy must stay below 2^31 granules (32-bit tile offsets)");
    if (workspace == nullptr || workspace_bytes < bwd_pair_workspace(cin, cout)) {
        set_error("conv2d_reflect_bwd_pair: workspace too small");
        return MMIF_EWORKSPACE;
    }
    return bwd_pair(make_tv(x), make_tv(gy), make_tv(gx), w_packed_t, dw, db, cin, cout, accumulate, (float*)workspace, (hipStream_t)stream);
}

// Backward of one WIDE 3x3 ConvLayer (Cin, Cout multiples of 64) as one call: the weight-gradient kernel also leaves the ReLU sign bytes of
// x in `signs`, the input-gradient kernel reads those instead of x.  Same results as mmif_conv2d_reflect_wgrad followed by
// mmif_conv2d_reflect_dgrad_folded(mask_bits, accum_bits = 0).
extern "C" int mmif_conv2d_bwd_wide_supported(int32_t cin, int32_t cout, int32_t ksize) { return bwd_wide_supported(ksize, cin, cout) ? 1 : 0; }
extern "C" size_t mmif_conv2d_bwd_wide_signs_bytes(int32_t n, int32_t cin, int32_t h, int32_t w) {   // (enough for either tensor dtype's sign layout)
    const size_t a = cin % 8 == 0 ? bwd_wide_signs_bytes(n, cin, h, w) : 0, b = x3_signs_bytes(n, (cin + 7) / 8, h, w);
    return a > b ? a : b;
}
// fp32 tensors: any 3x3 / 1x1 layer the split-operand kernels take (csrc/conv_x3.hip) -- the tensor-dependent conditions are checked by the call
extern "C" int mmif_conv2d_bwd_wide_supported_f32(int32_t cin, int32_t cout, int32_t ksize) {
    return (ksize == 3 || ksize == 1) && cin >= 1 && cout >= 1 && cin <= 512 ? 1 : 0;
}

extern "C" int mmif_conv2d_reflect_bwd_wide(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* x, const mmif_tensor* gx, float* dw,
                                            float* db, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits, int32_t accumulate,
                                            void* workspace, size_t workspace_bytes, void* signs, size_t signs_bytes, void* stream) {
    if (int rc = validate_tensor(gy, "gy")) return rc;
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    // accumulate: bit 0 = add onto dw / db; bits 1-2 = phase (0: both halves, 1: weight gradient + sign map only, 2: input gradient only,
    // reading the map phase 1 left) -- the two halves of one backward as two calls, so that a caller can time each kernel
    const int phase = (accumulate >> 1) & 3;
    accumulate &= 1;
    MMIF_REQUIRE(phase <= 2, "conv2d_reflect_bwd_wide: phase bits of `accumulate` must be 0, 1 or 2");
    if (gy->dtype == MMIF_F32 && x->dtype == MMIF_F32 && gx->dtype == MMIF_F32) {
        // fp32 tensors: the split-operand weight gradient leaves its sign map ([n][ceil(cb / 4)][h][w] dwords), the split-operand dgrad
        // masks with it, the stand-alone fold follows -- bit for bit mmif_conv2d_reflect_wgrad + mmif_conv2d_reflect_dgrad_folded
        MMIF_REQUIRE(w_packed_t != nullptr && dw != nullptr, "conv2d_reflect_bwd_wide: NULL operand image / dw");
        MMIF_REQUIRE(x->halo == 0 && gx->halo == 1, "conv2d_reflect_bwd_wide: x halo 0, gx halo 1 expected");
        MMIF_REQUIRE(gy->n == x->n && gy->h == x->h && gy->w == x->w && gx->n == x->n && gx->h == x->h && gx->w == x->w, "conv2d_reflect_bwd_wide: shape mismatch");
        MMIF_REQUIRE((cin + 7) / 8 == x->cb && x->cb == gx->cb && (cout + 7) / 8 == gy->cb, "conv2d_reflect_bwd_wide: channel blocks do not match");
        const TV tx = make_tv(x), tg = make_tv(gy), tgx = make_tv(gx);
        MMIF_REQUIRE(wgrad_x3_supported(ksize, cin, cout, tx, tg) && conv_x3_supported(true, ksize, cin, cout, tg, tgx),
                     "conv2d_reflect_bwd_wide: layer %d -> %d k%d / tensors not covered by the split-operand kernels", cin, cout, ksize);
        if (workspace == nullptr || workspace_bytes < wgrad_x3_workspace(cin, cout, ksize)) {
            set_error("conv2d_reflect_bwd_wide: workspace too small");
            return MMIF_EWORKSPACE;
        }
        if (signs == nullptr || signs_bytes < x3_signs_bytes(x->n, x->cb, x->h, x->w)) {
            set_error("conv2d_reflect_bwd_wide: sign-byte buffer too small (mmif_conv2d_bwd_wide_signs_bytes)");
            return MMIF_EWORKSPACE;
        }
        if (phase != 2)
            if (int rc = wgrad_x3(tx, tg, dw, db, cin, cout, accumulate, (float*)workspace, (hipStream_t)stream, ksize, (unsigned*)signs)) return rc;
        if (phase == 1) return MMIF_OK;
        bool folded = false;
        if (int rc = conv_x3(true, tg, tgx, tx, w_packed_t, nullptr, cin, cout, 0, mask_bits, 0, (hipStream_t)stream, ksize, (const unsigned*)signs, &folded)) return rc;
        return (ksize == 1 || folded) ? MMIF_OK : mmif_fold_halo(gx, stream);
    }
    MMIF_REQUIRE(bwd_wide_supported(ksize, cin, cout), "conv2d_reflect_bwd_wide: unsupported layer %d -> %d k%d", cin, cout, ksize);
    MMIF_REQUIRE(w_packed_t != nullptr && dw != nullptr, "conv2d_reflect_bwd_wide: NULL operand image / dw");
    MMIF_REQUIRE(gy->dtype == MMIF_BF16 && x->dtype == MMIF_BF16 && gx->dtype == MMIF_BF16, "conv2d_reflect_bwd_wide: bf16 tensors expected");
    MMIF_REQUIRE(gy->halo == 1 && (gy->flags & MMIF_T_FOLDED), "conv2d_reflect_bwd_wide: gy must be a folded halo-1 gradient");
    MMIF_REQUIRE(x->halo == 0 && gx->halo == 1, "conv2d_reflect_bwd_wide: x halo 0, gx halo 1 expected");
    MMIF_REQUIRE(gy->n == x->n && gy->h == x->h && gy->w == x->w && gx->n == x->n && gx->h == x->h && gx->w == x->w, "conv2d_reflect_bwd_wide: shape mismatch");
    MMIF_REQUIRE(cin / 8 == x->cb && x->cb == gx->cb && cout / 8 == gy->cb, "conv2d_reflect_bwd_wide: channel blocks do not match");
    MMIF_REQUIRE(x->h >= 4 && x->w >= 4, "conv2d_reflect_bwd_wide: needs h, w >= 4 (fold steps inside the border tiles)");
    MMIF_REQUIRE((long long)(x->h + 2) * (x->w + 2) * 16 * 8 < (1ll << 31), "conv2d_reflect_bwd_wide: 8 channel planes must stay below 2 GiB (32-bit staging offsets)");
    if (workspace == nullptr || workspace_bytes < wgrad_mfma_workspace(cin, cout, ksize)) {
        set_error("conv2d_reflect_bwd_wide: workspace too small");
        return MMIF_EWORKSPACE;
    }
    if (signs == nullptr || signs_bytes < bwd_wide_signs_bytes(x->n, cin, x->h, x->w)) {
        set_error("conv2d_reflect_bwd_wide: sign-byte buffer too small (mmif_conv2d_bwd_wide_signs_bytes)");
        return MMIF_EWORKSPACE;
    }
    return bwd_wide(make_tv(x), make_tv(gy), make_tv(gx), w_packed_t, dw, db, cin, cout, mask_bits, accumulate, (float*)workspace,
                    (unsigned char*)signs, (hipStream_t)stream, phase);
}
