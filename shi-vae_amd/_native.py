"""ctypes binding of libshivae_hip.so (C ABI: include/shivae_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHIVAE_LIB_PATH") or os.path.join(_HERE, "libshivae_hip.so")  # (override: experiment builds)

REAL, POS, BIN, CAT = 0, 1, 2, 3
TYPE_CODE = {"real": REAL, "pos": POS, "bin": BIN, "cat": CAT}

# parameter tensor ids (enum order of shivae_hip.h)
_P_NAMES = ["PHIX0_W", "PHIX0_B", "PHIX2_W", "PHIX2_B", "XS0_W", "XS0_B", "XS2_W", "XS2_B", "E0_W", "E0_B",
            "E2_W", "E2_B", "ENCMEAN_W", "ENCMEAN_B", "ENCSTD_W", "ENCSTD_B", "PHIZ_W", "PHIZ_B", "PRIOR0_W",
            "PRIOR0_B", "PRIORMEAN_W", "PRIORMEAN_B", "PRIORSTD_W", "PRIORSTD_B", "RNN_WIH", "RNN_WHH", "RNN_BIH",
            "RNN_BHH", "DEC0_W", "DEC0_B", "DEC2_W", "DEC2_B", "DEC4_W", "DEC4_B", "HEAD0_W", "HEAD0_B",
            "HEADMEAN_W", "HEADMEAN_B", "HEADSTD_W", "HEADSTD_B", "HEADOUT_W", "HEADOUT_B"]
P = {n: i for i, n in enumerate(_P_NAMES)}
_O_NAMES = ["Z", "Z_MEAN", "Z_STD", "ZP_MEAN", "ZP_STD", "S", "S_PROBS", "DEC_PARAMS", "ROWLOSS", "H_PREV", "PHI_Z",
            "A_S", "LOGITS", "E1", "E2", "GATES", "C_NEW", "D_GATES", "D_PHIZ_PRE", "D_MU", "D_STD_PRE", "D_E2_PRE",
            "D_E1_PRE", "D_LOGITS_PRE", "D_AS_PRE"]
O = {n: i for i, n in enumerate(_O_NAMES)}
STEP_FORWARD_ONLY, STEP_KEEP_PARAMS = 1, 2
SCALE_TRANSPOSE, SCALE_ZERO_FILL, SCALE_F64 = 1, 2, 4
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_RELU, EPI_SOFTPLUS, EPI_ADDC, EPI_MASK, EPI_ATOMIC = 1, 2, 4, 8, 16, 32

# every symbol include/shivae_hip.h declares
SYMBOLS = ["shivae_last_error", "shivae_version", "shivae_init", "shivae_plan_set_observed_fraction", "shivae_plan_create", "shivae_plan_destroy",
           "shivae_plan_num_params", "shivae_plan_param_info", "shivae_plan_x_dim", "shivae_plan_p_dim",
           "shivae_plan_chunk_rows", "shivae_plan_set_activation_rows", "shivae_plan_activation_rows", "shivae_plan_overflow_chunks",
           "shivae_plan_observed_fraction", "shivae_plan_workspace_bytes", "shivae_plan_bind", "shivae_elbo_step",
           "shivae_scan_fwd", "shivae_prior_kl", "shivae_decoder_nll", "shivae_scan_bwd", "shivae_loss_reduce",
           "shivae_set_gemm_split", "shivae_presplit_bytes", "shivae_presplit_weights", "shivae_gemm_nt_presplit", "shivae_sqnorm_noalloc", "shivae_comm_unique_id", "shivae_comm_init", "shivae_comm_destroy",
           "shivae_allreduce_grads",
           "shivae_sample_prior", "shivae_plan_set_noise_seed", "shivae_noise_fill", "shivae_get_output", "shivae_output_width", "shivae_get_persum", "shivae_heter_nll", "shivae_mc_impute", "shivae_kl_terms",
           "shivae_reparam_fwd", "shivae_reparam_bwd", "shivae_gemm", "shivae_zero", "shivae_sqnorm",
           "shivae_clip_adam", "shivae_scale_transform", "shivae_scale_inverse", "shivae_profile_start", "shivae_profile_active",
           "shivae_profile_stop", "shivae_debug_gemm_stamps"]


class Config(C.Structure):
    _fields_ = [("T", C.c_int), ("B", C.c_int), ("H", C.c_int), ("Z", C.c_int), ("K", C.c_int),
                ("n_attr", C.c_int), ("attr_type", C.POINTER(C.c_int)), ("attr_nclass", C.POINTER(C.c_int)),
                ("learn_std", C.c_int), ("chunk_rows", C.c_int)]


class ShiVAENativeError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises if it has not been built (run ``shi-vae_amd/build.sh``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ShiVAENativeError(
            "libshivae_hip.so not found at %s: build it with shi-vae_amd/build.sh (or __graft_entry__.build()). "
            "There is no CPU/PyTorch fallback for the Shi-VAE step." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ci, cf, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
    L.shivae_last_error.restype = C.c_char_p
    L.shivae_version.restype = C.c_char_p
    L.shivae_plan_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.shivae_plan_destroy.argtypes = [vp]
    L.shivae_plan_destroy.restype = None
    L.shivae_plan_num_params.argtypes = [vp]
    L.shivae_plan_num_params.restype = i64
    L.shivae_plan_param_info.argtypes = [vp, ci, ci, C.POINTER(i64), C.POINTER(ci), C.POINTER(ci)]
    for f in ("shivae_plan_x_dim", "shivae_plan_p_dim", "shivae_plan_chunk_rows", "shivae_plan_activation_rows"):
        getattr(L, f).argtypes = [vp]
    L.shivae_plan_set_activation_rows.argtypes = [vp, ci]
    L.shivae_plan_overflow_chunks.argtypes = [vp]
    L.shivae_plan_overflow_chunks.restype = C.c_longlong
    L.shivae_plan_observed_fraction.argtypes = [vp]
    L.shivae_plan_observed_fraction.restype = cf
    L.shivae_plan_set_observed_fraction.argtypes = [vp, cf]
    L.shivae_init.argtypes = []
    L.shivae_plan_workspace_bytes.argtypes = [vp]
    L.shivae_plan_workspace_bytes.restype = C.c_size_t
    L.shivae_plan_bind.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.shivae_elbo_step.argtypes = [vp, vp, vp, vp, vp, cf, cf, cf, ci, vp, vp]
    L.shivae_scan_fwd.argtypes = [vp, vp, vp, vp, cf, vp]
    L.shivae_prior_kl.argtypes = [vp, cf, cf, ci, vp]
    L.shivae_decoder_nll.argtypes = [vp, vp, vp, cf, ci, vp]
    L.shivae_scan_bwd.argtypes = [vp, vp, vp, vp, cf, vp, vp, vp, vp, vp]
    L.shivae_loss_reduce.argtypes = [vp, cf, cf, vp, vp]
    L.shivae_set_gemm_split.argtypes = [ci]
    L.shivae_presplit_bytes.argtypes = [ci, ci]
    L.shivae_presplit_bytes.restype = C.c_size_t
    L.shivae_presplit_weights.argtypes = [vp, ci, ci, ci, vp, vp]
    L.shivae_gemm_nt_presplit.argtypes = [ci, ci, ci, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, vp]
    L.shivae_sqnorm_noalloc.argtypes = [vp, i64, vp, vp]
    L.shivae_comm_unique_id.argtypes = [vp]
    L.shivae_comm_init.argtypes = [C.POINTER(vp), ci, ci, vp]
    L.shivae_comm_destroy.argtypes = [vp]
    L.shivae_allreduce_grads.argtypes = [vp, i64, vp, i64, vp]
    L.shivae_sample_prior.argtypes = [vp, vp, vp, cf, vp]
    L.shivae_plan_set_noise_seed.argtypes = [vp, C.c_uint64]
    L.shivae_noise_fill.argtypes = [C.c_uint64, i64, ci, ci, vp, vp, vp]
    L.shivae_get_output.argtypes = [vp, ci, vp, vp]
    L.shivae_output_width.argtypes = [vp, ci]
    L.shivae_get_persum.argtypes = [vp, vp, vp]
    L.shivae_heter_nll.argtypes = [vp, vp, vp, vp, ci, vp, ci, cf, vp, vp]
    L.shivae_mc_impute.argtypes = [vp, vp, ci, ci, ci, C.c_uint64, vp, vp]
    L.shivae_kl_terms.argtypes = [vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, vp, vp]
    L.shivae_reparam_fwd.argtypes = [vp, vp, ci, ci, cf, vp, vp, vp, vp, vp, ci, ci, vp, ci, vp]
    L.shivae_reparam_bwd.argtypes = [vp, vp, ci, ci, cf, vp, vp, vp, ci, ci, vp, vp, ci, vp]
    L.shivae_gemm.argtypes = [ci, ci, ci, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, vp]
    L.shivae_zero.argtypes = [vp, i64, vp]
    L.shivae_sqnorm.argtypes = [vp, i64, vp, vp, vp]
    L.shivae_clip_adam.argtypes = [vp, vp, vp, vp, i64, vp, cf, cf, cf, cf, cf, ci, vp]
    L.shivae_scale_transform.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp]
    L.shivae_scale_inverse.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, vp]
    L.shivae_debug_gemm_stamps.argtypes = [vp]
    L.shivae_profile_start.argtypes = []
    L.shivae_profile_active.argtypes = []
    L.shivae_profile_stop.argtypes = [C.POINTER(C.c_double), ci]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise ShiVAENativeError("libshivae_hip: %s (code %d)" % (lib().shivae_last_error().decode(), rc))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """Owns one shivae_plan handle."""

    def __init__(self, T, B, H, Z, K, attr_types, attr_nclass, learn_std, chunk_rows=0, act_rows=0):
        n = len(attr_types)
        self._types = (C.c_int * n)(*attr_types)
        self._ncls = (C.c_int * n)(*attr_nclass)
        cfg = Config(T, B, H, Z, K, n, self._types, self._ncls, int(bool(learn_std)), int(chunk_rows))
        h = C.c_void_p()
        check(lib().shivae_plan_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.T, self.B = T, B
        if act_rows:  # (rows of the decoder's activation buffers, when fewer than the chunk's: include/shivae_hip.h)
            check(lib().shivae_plan_set_activation_rows(self.h, int(act_rows)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().shivae_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def num_params(self):
        return int(lib().shivae_plan_num_params(self.h))

    @property
    def x_dim(self):
        return int(lib().shivae_plan_x_dim(self.h))

    @property
    def p_dim(self):
        return int(lib().shivae_plan_p_dim(self.h))

    @property
    def chunk_rows(self):
        return int(lib().shivae_plan_chunk_rows(self.h))

    @property
    def activation_rows(self):
        return int(lib().shivae_plan_activation_rows(self.h))

    @property
    def overflow_chunks(self):
        return int(lib().shivae_plan_overflow_chunks(self.h))

    @property
    def observed_fraction(self):
        return float(lib().shivae_plan_observed_fraction(self.h))

    @property
    def workspace_bytes(self):
        return int(lib().shivae_plan_workspace_bytes(self.h))

    def param_info(self, tensor_id, attr=0):
        off, r, c = C.c_int64(), C.c_int(), C.c_int()
        rc = lib().shivae_plan_param_info(self.h, tensor_id, attr, C.byref(off), C.byref(r), C.byref(c))
        if rc != 0:
            return None
        return off.value, r.value, c.value

    def bind(self, workspace, params, grads):
        check(lib().shivae_plan_bind(self.h, ptr(workspace), workspace.numel() * workspace.element_size(),
                                     ptr(params), ptr(grads)))

    def output_width(self, which):
        return int(lib().shivae_output_width(self.h, which))


_TILES = ["256x32", "32x256", "256x64", "64x256", "128x128", "128x128 split-bf16"]
_LAYOUTS = ["NT", "NN", "TN"]


def profile_name(i):
    i = int(i)
    if i < 18:
        return "gemm_kernel<%s,%s>" % (_LAYOUTS[i // 6], _TILES[i % 6])
    return {18: "scan_fwd_kernel", 19: "scan_bwd_kernel", 20: "heter_nll_kernel", 21: "kl_kernel", 22: "head_fwd_kernel",
            23: "head_bwd_kernel", 24: "gemm_split_nt_kernel<128x256, pre-split W>", 25: "gemm_split_tn_kernel<128x256x16>"}[i]


def profile_start():
    check(lib().shivae_profile_start())


def profile_stop():
    """-> list of dicts {id, name, launches, ms, work} (work: flops, or bytes for the loss kernels)."""
    buf = (C.c_double * (4 * 32))()
    n = lib().shivae_profile_stop(buf, 32)
    if n < 0:
        check(n)
    return [dict(id=int(buf[4 * i]), name=profile_name(buf[4 * i]), launches=int(buf[4 * i + 1]), ms=buf[4 * i + 2],
                 work=buf[4 * i + 3]) for i in range(n)]
