"""Compile-time instances stand for FAMILIES of configurations (csrc/aie_layout.h: aie_spec_normalize): which
configurations fall into a family is decided on the host, by comparing normalised parameter blocks -- checked here
without a GPU, through the build-time tool that makes the instances' images (csrc/aie_specgen.c)."""
import ctypes
import subprocess

import pytest

from helpers import FAMILY_CASES, make_env


def _normalised_block(cfg):
    from ai_economist_amd import _specs

    env = make_env(cfg, n_envs=3)
    c = env.build_config()
    raw = bytes(ctypes.string_at(ctypes.addressof(c), ctypes.sizeof(c)))
    return subprocess.run([_specs._tool()], input=raw, check=True, capture_output=True).stdout


@pytest.fixture(scope="module")
def images():
    from ai_economist_amd import _specs

    return [img for _name, img, _family, _waves in _specs.images()]


@pytest.mark.parametrize("case", sorted(FAMILY_CASES))
def test_family_membership(case, images):
    cfg, in_family = FAMILY_CASES[case]
    block = _normalised_block(cfg)
    hits = [k for k, img in enumerate(images) if img == block]
    assert (len(hits) == 1) == in_family, "%s: matches instances %s" % (case, hits)


def test_generated_header_is_current(images):
    """csrc/aie_spec_generated.h (committed) holds exactly the images the current layout code produces."""
    from ai_economist_amd import _specs

    assert open(_specs.OUT).read() == _specs.header_text()


def test_blanked_scalars_are_poisoned_not_zero(images):
    """A kernel line that read a blanked scalar from the image would get 0x5A5A5A5A, not a plausible 0."""
    from ai_economist_amd import _cabi

    cfg_t = _cabi.AieConfig
    img = images[0]
    for field in ("episode_length", "build_payment", "cda_order_duration", "tax_period"):
        off = getattr(cfg_t, field).offset
        assert img[off:off + 4] == b"\x5a" * 4, field
    for field in ("starting_agent_coin", "isoelastic_eta", "energy_cost", "move_labor", "tax_rate_max"):
        off = getattr(cfg_t, field).offset
        assert img[off:off + 8] == b"\x5a" * 8, field
    off = cfg_t.n_agents.offset
    assert int.from_bytes(img[off:off + 4], "little") == 4  # what shapes the code stays
