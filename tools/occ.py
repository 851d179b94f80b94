import ctypes, sys
L=ctypes.CDLL('/root/repo/xmm-superres-denoise_amd/lib/libxsd_hip.so')
L.xsd_debug_residency_ms.restype=ctypes.c_float
for lds in (0, 65536, 80384, 81024, 81920):
    print('lds',lds,'API blocks/CU',L.xsd_debug_occupancy(lds), 'sleep-test ms: grid256 %.2f grid512 %.2f grid768 %.2f grid1024 %.2f'%tuple(L.xsd_debug_residency_ms(g,256,lds,1000) for g in (256,512,768,1024)))
