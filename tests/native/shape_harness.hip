// Host-only harness for the launch-shape functions of the marching kernels (no GPU needed: without a device the capacity falls
// back to 256 CUs x 12 waves). Prints, per grid, the segment table wet_launch_shape() builds: tests/test_launch_shape_cpu.py checks
// that every row is covered exactly once in every band and that no segment is empty.
#include "../../2d-weather-sandbox_amd/csrc/wx_wet.h"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
  printf("[");
  for (int a = 1; a + 1 < argc; a += 2) {
    wx::Geo g{};
    g.X = atoi(argv[a]);
    g.Y = atoi(argv[a + 1]);
    const wx::WetLaunch w = wx::wet_launch_shape(g, -1);
    printf("%s{\"X\": %d, \"Y\": %d, \"n_strips\": %d, \"bands\": %d, \"n_seg\": %d, \"start\": [", a > 1 ? ", " : "", g.X,
           g.Y, w.n_strips, w.segs.bands, w.segs.n_seg);
    for (int s = 0; s <= w.segs.n_seg; s++) printf("%s%d", s ? ", " : "", w.segs.start[s]);
    const wx::WetLaunch hv = wx::wet_shape_halved(w); // (what the edge strips of a slab run with, next to the interior strips)
    printf("], \"halved_n_seg\": %d, \"halved_start\": [", hv.segs.n_seg);
    for (int s = 0; s <= hv.segs.n_seg; s++) printf("%s%d", s ? ", " : "", hv.segs.start[s]);
    printf("]}");
  }
  printf("]\n");
  return 0;
}
