#!/bin/bash
# Pins the last unpinned piece of the visual path -- the iterations inside ceres::Solve -- on a machine that HAS the reference's
# dependencies (Ceres Solver 2.1.0, Eigen 3.3.7, and what include/utils.hpp pulls in: OpenCV, Sophus, PCL).  Neither the build
# image nor the GPU box has them (probe: gpurun_out/r4a/ceres_probe.txt), so this cannot run there.
#   tools/pin_ceres/pin_ceres.sh            builds tools/pin_ceres/ceres_pin_driver and runs tests/test_gpu_ceres_pin.py
#   LVBA_REFERENCE=/path/to/Global-LVBA     the reference checkout whose include/utils.hpp supplies the cost functors
# The test then compares lvba_visual_refine's per-iteration trace (cost, trust-region radius, accepted flag) and the refined
# cameras with the real solver's, to north_star's 1e-5.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
REF=${LVBA_REFERENCE:-/root/reference}
[ -f "$REF/include/utils.hpp" ] || { echo "pin_ceres: $REF/include/utils.hpp not found (set LVBA_REFERENCE)"; exit 3; }
CFLAGS=$(pkg-config --cflags eigen3 2>/dev/null || echo "-I/usr/include/eigen3")
CFLAGS="$CFLAGS $(pkg-config --cflags opencv4 2>/dev/null || pkg-config --cflags opencv 2>/dev/null || true)"
CFLAGS="$CFLAGS $(pkg-config --cflags pcl_common 2>/dev/null || ls -d /usr/include/pcl-* 2>/dev/null | head -1 | sed 's/^/-I/')"
LIBS="-lceres -lglog $(pkg-config --libs opencv4 2>/dev/null || pkg-config --libs opencv 2>/dev/null || true)"
echo '#include <ceres/version.h>
#if CERES_VERSION_MAJOR != 2 || CERES_VERSION_MINOR != 1
#warning "the reference pins Ceres Solver 2.1.0 (README.md:20); another version is installed"
#endif
int main(){return 0;}' > /tmp/lvba_ceres_probe.cpp
g++ -std=c++17 $CFLAGS /tmp/lvba_ceres_probe.cpp -o /tmp/lvba_ceres_probe || { echo "pin_ceres: no usable Ceres installation (ceres/version.h)"; exit 4; }
g++ -O2 -std=c++17 -I"$REF/include" $CFLAGS "$HERE/ceres_pin_driver.cpp" -o "$HERE/ceres_pin_driver" $LIBS -lpthread
echo "built $HERE/ceres_pin_driver"
cd "$ROOT" && LVBA_CERES_PIN_DRIVER="$HERE/ceres_pin_driver" python -m pytest tests/test_gpu_ceres_pin.py -q -m gpu -p no:cacheprovider "$@"
