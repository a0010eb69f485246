"""The oracle's restatement of the reference's compute shader (Compute.shader:140-345, `orc_render_glsl`) against its
restatement of the CPU loop (Source.cpp:530-628): the two renderers of the reference trace the same rays and differ
only in how a lit pixel's colour is computed.  PARITY UNPINNED for this mode: no GL here to run the shader itself."""
import numpy as np

from raytracing_amd import scenes


def _both(oracle, desc):
    tris, lights = oracle.build_scene(desc)
    tr = oracle.camera_trans(desc.cam_rot)
    cpu = oracle.render(tris, lights, desc.cam_pos, tr, desc.width, desc.height, desc.depth)
    glsl = oracle.render_glsl(tris, lights, desc.cam_pos, tr, desc.width, desc.height)
    return tris, lights, tr, cpu, glsl


def test_matte_scene_same_pixels_lit_and_colours_within_rounding(oracle):
    desc = scenes.stock(120, 90)
    _, _, _, cpu, glsl = _both(oracle, desc)
    assert np.array_equal(cpu > 0, glsl > 0)                      # same hits, same shadows
    below = cpu < 1.0                                             # Source.cpp:602-610 clamps, the shader does not
    assert below.mean() > 0.5
    rel = np.abs(glsl[below] - cpu[below]) / np.maximum(cpu[below], 1e-30)
    assert rel.max() < 4 * 2.0 ** -23                            # three roundings against one, nothing more
    assert (glsl[~below] >= 1.0).all()
    assert (glsl != cpu).any()                                    # ... and it is not the same arithmetic


def test_shader_has_no_mirrors(oracle):
    desc = scenes.room(96, 96)
    tris, lights, tr, cpu, glsl = _both(oracle, desc)
    assert (tris[:, 27] > 0.5).any()
    matte = tris.copy()
    matte[:, 27] = 0.0
    again = oracle.render_glsl(matte, lights, desc.cam_pos, tr, desc.width, desc.height)
    assert np.array_equal(glsl.view(np.uint32), again.view(np.uint32))      # shininess is not read
    assert (glsl != cpu).mean() > 0.05                                     # the mirror walls show their own colour


def test_lights_add_up_before_the_colour(oracle):
    desc = scenes.two_lights(64, 64)
    tris, lights, tr, cpu, glsl = _both(oracle, desc)
    one = oracle.render_glsl(tris, lights[:1], desc.cam_pos, tr, 64, 64)
    two = oracle.render_glsl(tris, lights[1:], desc.cam_pos, tr, 64, 64)
    assert np.allclose(glsl, one + two, rtol=1e-6, atol=0)
    assert (glsl > np.maximum(one, two)).any()
    none = oracle.render_glsl(tris, lights[:0], desc.cam_pos, tr, 64, 64)
    assert not none.any()
