// ref_esti_plane.cpp -- harness around the REFERENCE's esti_plane<float> (include/common_lib.h:225-257), which
// build_ref.sh extracts from the reference tree into oracle/_ref/esti_plane_ref.inc at build time.  TEST INFRASTRUCTURE.
// Only the few names that function needs from common_lib.h are declared here (the header itself pulls in PCL and ROS).
//   in : uint32 n, then n x 15 float (5 points x xyz)
//   out: uint32 n, then n x {4 float pabcd, uint32 ok}
#include <Eigen/Dense>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

using namespace std;
using namespace Eigen;
#define NUM_MATCH_POINTS (5)  // include/common_lib.h:26
struct PointType { float x, y, z; };  // the members esti_plane reads of pcl::PointXYZINormal (include/common_lib.h:37)
typedef vector<PointType, Eigen::aligned_allocator<PointType>> PointVector;  // :39

#include "esti_plane_ref.inc"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    vector<float> in((size_t)n * 15);
    if (fread(in.data(), 4, in.size(), f) != in.size()) return 2;
    fclose(f);
    FILE* o = fopen(argv[2], "wb");
    fwrite(&n, 4, 1, o);
    for (uint32_t i = 0; i < n; ++i) {
        PointVector pts(NUM_MATCH_POINTS);
        for (int j = 0; j < NUM_MATCH_POINTS; ++j) { pts[j].x = in[i * 15 + 3 * j]; pts[j].y = in[i * 15 + 3 * j + 1]; pts[j].z = in[i * 15 + 3 * j + 2]; }
        Matrix<float, 4, 1> pabcd;
        const bool ok = esti_plane(pabcd, pts, 0.1f);  // the call at src/laserMapping.cpp:678
        float out[4] = {pabcd(0), pabcd(1), pabcd(2), pabcd(3)};
        uint32_t k = ok ? 1u : 0u;
        fwrite(out, 4, 4, o);
        fwrite(&k, 4, 1, o);
    }
    fclose(o);
    printf("ref_esti_plane: %u fits, Eigen %d.%d.%d, SSE2 %s\n", n, EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION,
#ifdef EIGEN_VECTORIZE_SSE2
           "on"
#else
           "off"
#endif
    );
    return 0;
}
