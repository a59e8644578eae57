// TEST INFRASTRUCTURE: C entry for the g++ emulation build of stagei.hip (tests/emu/build_emu.py).
#include "../../moshpp_amd/csrc/stagei_views.h"
#include "../../include/moshii.h"

extern "C" int stagei_emu_solve(const S1ModelView* mv, const S1PriorView* pv, const moshii_stagei_desc* ds, char* err, int errlen) {
    return moshii_stagei_core(mv, pv, ds, nullptr, err, errlen);
}
