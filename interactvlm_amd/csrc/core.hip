// Library identity + error strings for libivlm_hip.so.
#include "ivlm_common.h"

static thread_local char g_last_hip_error[512] = "";

extern "C" {

void ivlm_set_last_hip_error(int code, const char* where) {
    snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s (hipError %d) at %s",
             hipGetErrorString((hipError_t)code), code, where ? where : "?");
}

const char* ivlm_last_hip_error(void) { return g_last_hip_error; }

int ivlm_abi_version(void) { return 2; }  // 2: fp32 residual streams / fp32 activation flags (round 2)

const char* ivlm_build_arch(void) { return "gfx950"; }

const char* ivlm_error_string(int code) {
    switch (code) {
        case IVLM_OK: return "ok";
        case IVLM_ERR_INVALID_ARG: return "invalid argument";
        case IVLM_ERR_WORKSPACE: return "workspace too small";
        case IVLM_ERR_LAUNCH: return "kernel launch / HIP runtime error";
        case IVLM_ERR_UNSUPPORTED: return "unsupported dtype or configuration";
        default: return "unknown error";
    }
}

}  // extern "C"
