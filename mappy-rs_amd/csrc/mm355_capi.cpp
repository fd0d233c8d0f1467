// mm355_capi.cpp -- small C-ABI helpers (error strings, version)
#include "mm355_host.h"

extern "C" const char *mm355_strerror(int code)
{
	switch (code) {
	case MM355_OK: return "ok";
	case MM355_ENODEV: return "no gfx950 HIP device is visible (libmm355 has no CPU fallback)";
	case MM355_EINVAL: return "invalid argument";
	case MM355_ENOMEM: return "out of memory or capacity exceeded";
	case MM355_EIO: return "Did not create or open an index";
	case MM355_ENOIDX: return "No index";
	case MM355_EEMPTY: return "Sequence is empty";
	case MM355_EUNSUP: return "option outside the long-read hot path (sr/splice presets, query-strand / heap-sort flags)";
	case MM355_EHIP: return "a HIP runtime call failed";
	default: return "unknown error";
	}
}

extern "C" const char *mm355_version(void) { return "mm355 0.1 (minimap2 2.26 semantics; gfx950)"; }
