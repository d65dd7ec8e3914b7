// Stand-in for <bgfx/embedded_shader.h>; the macros live in the stub's bgfx.h (shader blobs are never used).
#include "bgfx.h"
