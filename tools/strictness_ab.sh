#!/bin/bash
# What the arithmetic strictness of the kernels costs: the SAME sources
# built with what the reference's own build options allow (core.clj:128 :fast-math :enable-mad) --
# contraction, approximate divide/sqrt, fast-math -- timed and scored with BASELINE's metric against
# the reference kernel built with those options for this chip (`fast`).  Never the product.
#   build container:  tools/strictness_ab.sh build      GPU box:  tools/strictness_ab.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
case "$1" in
  build)
    python tools/ab_build.py contract="-ffp-contract=fast" apxdiv="-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt" \
      fastmath="-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -ffast-math" ;;
  run)
    mkdir -p gpurun_out/strict
    P='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("   kernel ms: gfx950 contract %.3f, cpu contract %.3f" % (j["roofline"]["kernel_ms"], j["other_contract"]["kernel_ms"]))'
    for v in product contract apxdiv fastmath; do
      echo "== $v"
      if [ $v = product ]; then unset RAYMARCH_LIB; else export RAYMARCH_LIB=libraymarch_hip_ab_$v.so; fi
      python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "$P"
      python tools/pin_gfx950.py --quick --out gpurun_out/strict/pin_$v.txt 2>/dev/null | grep "contract\|cast" | sed 's/^/  /'
    done ;;
  *) echo "usage: $0 build|run"; exit 2 ;;
esac
