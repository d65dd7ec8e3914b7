# vgx_tessellate on a million distinct stroked cubics: the one-walk route against the heap route, same box
for r in 1 2; do for v in 1 0; do echo "VGX_TESS_FLAT1=$v"; VGX_TESS_FLAT1=$v timeout 300 python profiles/experiments/r06_cubics_tessellate.py 2>&1 | tail -1; done; done
