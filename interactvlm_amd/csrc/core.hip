// Library identity + error strings for libivlm_hip.so.
#include "ivlm_common.h"

extern "C" {

int ivlm_abi_version(void) { return 1; }

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
