// node_lines.cpp -- compile check of the drop-in boundary: the lines of the reference's node that touch the filter,
// AS THEY STAND in src/laserMapping.cpp, against the mirrored headers.  What a maintainer changes in the node is the
// include block and one assignment (the handle); the registration (:828) and the update (:960-961) stay as written.
//
//   g++ -std=c++17 -Iinclude examples/node_lines.cpp -Lfast_lio_amd/lib -lfastlio_hip -Wl,-rpath,$PWD/fast_lio_amd/lib
#include <algorithm>
#include <cstdio>

#include "fastlio_amd/esekfom.hpp"
#include "fastlio_amd/h_share_model.hpp"
#include "fastlio_amd/use-ikfom.hpp"
#include "fastlio_hip.h"

using namespace std;
using fastlio_amd::h_share_model;  // replaces the node's own h_share_model (src/laserMapping.cpp:638-754)

#define NUM_MAX_ITERATIONS_DEFAULT 4
#define LASER_POINT_COV (0.001)                             // src/laserMapping.cpp:64
int NUM_MAX_ITERATIONS = NUM_MAX_ITERATIONS_DEFAULT;        // :71 (set from the "max_iteration" parameter, :768)
double solve_H_time = 0;                                    // :65
esekfom::esekf<state_ikfom, 12, input_ikfom> kf;            // :130
state_ikfom state_point;                                    // :131

int main() {
    flh_handle* handle = nullptr;
    if (flh_create(nullptr, &handle) != 0) {
        printf("flh_create: %s\n", flh_last_error());
        return 2;
    }
    fastlio_amd::g_hshare.handle = handle;  // the one added line: what ikdtree + the scan globals were to the old h_share_model

    // ---- src/laserMapping.cpp:826-828, verbatim
    double epsi[23] = {0.001};
    fill(epsi, epsi+23, 0.001);
    kf.init_dyn_share(get_f, df_dx, df_dw, h_share_model, NUM_MAX_ITERATIONS, epsi);

    // ---- src/laserMapping.cpp:960-961, verbatim (needs a map and a scan on the handle to do anything useful)
    if (flh_map_size(handle) > 0 && flh_scan_size(handle) > 0) {
    kf.update_iterated_dyn_share_modified(LASER_POINT_COV, solve_H_time);
    state_point = kf.get_x();
    }
    printf("node lines compiled and registered (pos %.1f)\n", state_point.pos[0]);
    flh_destroy(handle);
    return 0;
}
