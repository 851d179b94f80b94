import ctypes, sys
L=ctypes.CDLL('/root/repo/xmm-superres-denoise_amd/lib/libxsd_hip.so')
for lds in (0, 32768, 49152, 65536, 66000, 70000, 75000, 78000, 79000, 80000, 80384, 81024, 81920, 82000):
    print(lds, L.xsd_debug_occupancy(lds))
