import ctypes as C, sys
sys.path.insert(0, '/root/repo')
from gsplat_amd import capi
L = capi.lib()
for mb in (256, 1024, 4096):
    for rep in (3,):
        g = C.c_double()
        rc = L.gs_debug_hbm_copy_rate(mb << 20, rep, C.byref(g))
        print(mb, 'MiB', rc, round(g.value, 1), 'GB/s form', L.gs_debug_hbm_copy_form(), flush=True)
