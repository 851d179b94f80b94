"""Host-side inference plumbing (no GPU): FITS primary-HDU round trip and the WCS header arithmetic of the reference's
write_xmm_file_to_fits_wcs (utils/filehandling.py:199-226): CRPIX + (6, 2) for the centred pad; for the 2x model
CRPIX -> 2*CRPIX + 0.5, CDELT / 2 and a CD matrix rotated by 90 - PA_PNT."""
import math
import os
from collections import OrderedDict

import numpy as np


def test_fits_round_trip_and_wcs(tmp_path):
    from xmm_superres_denoise.infer import read_fits, wcs_header, write_fits
    hdr = OrderedDict(CRPIX1=201.5, CRPIX2=205.5, CDELT1=-0.0011, CDELT2=0.0011, CTYPE1="RA---TAN", PA_PNT=37.25,
                      EXPOSURE=20000.0, LTV1=3.0, OBS_ID="0123456789")
    img = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.25
    p = os.path.join(tmp_path, "a.fits.gz")
    write_fits(p, img, wcs_header(hdr, "src.fits", 1, 20000.0))
    back, h = read_fits(p)
    assert np.array_equal(back.astype(np.float32), img)
    assert h["CRPIX1"] == 207.5 and h["CRPIX2"] == 207.5 and "LTV1" not in h and h["OBS_ID"] == "0123456789"
    assert h["IMG_FILE"] == "src.fits" and h["EXPOSURE"] == 20000.0
    h2 = wcs_header(hdr, "src.fits", 2, 20000.0)
    assert h2["CRPIX1"] == 2 * 207.5 + 0.5 and h2["CRPIX2"] == 2 * 207.5 + 0.5
    assert h2["CDELT1"] == -0.00055 and h2["CDELT2"] == 0.00055
    r = math.radians(90.0 - 37.25)
    assert abs(h2["CD1_1"] - (-0.00055 * math.cos(r))) < 1e-15 and abs(h2["CD1_2"] - (-0.00055 * math.sin(r))) < 1e-15
    assert abs(h2["CD2_1"] - (-0.00055 * math.sin(r))) < 1e-15 and abs(h2["CD2_2"] - (0.00055 * math.cos(r))) < 1e-15
    assert h2["CROT2"] == 90.0 - 37.25
