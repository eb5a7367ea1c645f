#!/usr/bin/env python3
"""Turns the reference's own rendering of its glassWolf scene into a data fixture. TEST INFRASTRUCTURE.

CW/raytracer/screenshot.bmp is what the reference wrote with SDL_SaveImage (main.cpp:105) after
rendering the scene of main.cpp:78-89 at 500 x 500 (definitions.h:43-44) for about 10^4 samples per
pixel (Screenshots/glasswolf_10000SPP_500P.png is a capture of the same window): the ARGB8888
screen buffer that Draw's tonemap loop (main.cpp:141-145, SDLauxiliary.h:149-161) filled. It is the
only image-level output of the path that the reference's files hold, and it pins what the call
counts cannot: the radiance values themselves (light power, bounce weights, Fresnel, tonemap).

    python oracle/ref_harness/screenshot_fixture.py   ->  tests/golden/glasswolf_screen_500x500.npz

Reads /root/reference at fixture-generation time only; the fixture holds pixels, no source.
"""
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("REF", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    im = Image.open(os.path.join(REF, "CW", "raytracer", "screenshot.bmp"))
    a = np.asarray(im.convert("RGBA"))
    assert a.shape == (500, 500, 4), a.shape
    assert (a[..., 3] == 128).all()  # PutPixelSDL's alpha (SDLauxiliary.h:160)
    rgb = np.ascontiguousarray(a[..., :3])  # [row y][column x][r, g, b]
    out = os.path.join(ROOT, "tests", "golden", "glasswolf_screen_500x500.npz")
    np.savez_compressed(out, rgb=rgb)
    print(out, rgb.shape, rgb.reshape(-1, 3).mean(0), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
