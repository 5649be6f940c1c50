// local_map.hpp -- the local-map cube that follows the LiDAR (reference: lasermap_fov_segment, src/laserMapping.cpp:230-280).
// Pure host scalar code: it only decides WHICH axis-aligned slabs of the map fall out of the cube; the points are
// removed on the device by flh_map_delete_boxes (ikdtree.Delete_Point_Boxes, :275).
// Types follow the reference's declarations: float box corners (BoxPointType), float DET_RANGE / MOV_THRESHOLD
// (:77-78), double cube_len (:91), so the slabs come out bit-identical.
#pragma once
#include <cmath>
#include <vector>

namespace fastlio_amd {

struct BoxPointType {
    float vertex_min[3];
    float vertex_max[3];
};

struct LocalMap {
    BoxPointType LocalMap_Points{};
    bool Localmap_Initialized = false;
    double cube_len = 200.0;       // cube_side_length (:774)
    float DET_RANGE = 300.0f;      // mapping/det_range (:775)
    static constexpr float MOV_THRESHOLD = 1.5f;

    // Returns the slabs to delete (cub_needrm); empty when the cube did not move.
    std::vector<BoxPointType> lasermap_fov_segment(const double pos_LiD[3]) {
        std::vector<BoxPointType> cub_needrm;
        if (!Localmap_Initialized) {
            for (int a = 0; a < 3; ++a) {
                LocalMap_Points.vertex_min[a] = (float)(pos_LiD[a] - cube_len / 2.0);
                LocalMap_Points.vertex_max[a] = (float)(pos_LiD[a] + cube_len / 2.0);
            }
            Localmap_Initialized = true;
            return cub_needrm;
        }
        const float trigger = MOV_THRESHOLD * DET_RANGE;
        float lo_gap[3], hi_gap[3];
        bool need_move = false;
        for (int a = 0; a < 3; ++a) {
            lo_gap[a] = (float)std::fabs(pos_LiD[a] - (double)LocalMap_Points.vertex_min[a]);
            hi_gap[a] = (float)std::fabs(pos_LiD[a] - (double)LocalMap_Points.vertex_max[a]);
            need_move = need_move || lo_gap[a] <= trigger || hi_gap[a] <= trigger;
        }
        if (!need_move) return cub_needrm;
        const double by_cube = (cube_len - 2.0 * MOV_THRESHOLD * DET_RANGE) * 0.5 * 0.9;
        const double by_range = (double)(DET_RANGE * (MOV_THRESHOLD - 1));
        const float mov_dist = (float)(by_cube > by_range ? by_cube : by_range);
        const BoxPointType old = LocalMap_Points;
        BoxPointType moved = old;
        for (int a = 0; a < 3; ++a) {
            BoxPointType slab = old;
            if (lo_gap[a] <= trigger) {          // close to the low face: shift down, drop the top slab
                moved.vertex_max[a] -= mov_dist;
                moved.vertex_min[a] -= mov_dist;
                slab.vertex_min[a] = old.vertex_max[a] - mov_dist;
                cub_needrm.push_back(slab);
            } else if (hi_gap[a] <= trigger) {   // close to the high face: shift up, drop the bottom slab
                moved.vertex_max[a] += mov_dist;
                moved.vertex_min[a] += mov_dist;
                slab.vertex_max[a] = old.vertex_min[a] + mov_dist;
                cub_needrm.push_back(slab);
            }
        }
        LocalMap_Points = moved;
        return cub_needrm;
    }
};

}  // namespace fastlio_amd
