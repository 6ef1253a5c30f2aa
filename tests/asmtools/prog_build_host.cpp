// Host build of the program builder (csrc/cbca_prog_build.h) for the CPU tests: the same function the device runs,
// compiled with g++ against a layout header generated for the test's parameters, compared word for word with the
// plain-Python statement (tests/asmtools/cbca_prog_ref.py).   g++ -O1 -shared -fPIC -I<csrc> -I<dir of layout.h>
#include "cbca_prog_build.h"
#include "layout.h"          // generated: defines TEST_LAYOUT (a CBCA_PROG_V*_LAYOUT macro)

static const mccnn::prog::Layout kL = TEST_LAYOUT;

extern "C" int prog_build_patch(const uint32_t *sup0, int H, int W, int y0, int x0, uint32_t *out, int cap)
{
    return mccnn::prog::build_patch(kL, sup0, H, W, y0, x0, out, cap);
}
extern "C" int prog_build_patch_skip(const uint32_t *sup0, int H, int W, int y0, int x0, uint32_t *out, int cap)
{
    return mccnn::prog::build_patch(kL, sup0, H, W, y0, x0, out, cap, true);
}
extern "C" int prog_stride_dwords(void) { return mccnn::prog::stride_dwords(kL.K, kL.G, kL.W); }
extern "C" int prog_band_rows(int H) { return mccnn::prog::band_rows_of(H, kL.K); }
