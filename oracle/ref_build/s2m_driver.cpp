// TEST INFRASTRUCTURE.  C entry over the reference's closed-form point-to-triangle distance derivatives, compiled from the reference's
// own header where it lies (-I /root/reference/src/moshpp/scan2mesh/mesh_distance): sample2meshdist.h:67-205.
#include "sample2meshdist.h"

// squared distance of x to the `part` (0 plane, 1-3 edges ab/bc/ca, 4-6 vertices a/b/c) of triangle abc, and its gradients
extern "C" double s2m_ref_squared(int part, const double* x, const double* a, const double* b, const double* c,
                                  double* dx, double* da, double* db, double* dc) {
    for (int i = 0; i < 3; ++i) dx[i] = da[i] = db[i] = dc[i] = 0.0;
    instances::SquaredDistance d;
    return d.tri(part, x, a, b, c, dx, da, db, dc);
}
