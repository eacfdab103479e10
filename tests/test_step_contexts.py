"""Host-side switches of the train step (no GPU): the contexts the engine wraps around the forward / loss."""
from immunostruct_amd import functional as HF


def test_speculative_backward_context_toggles_and_restores():
    assert HF.SpeculativeBackward.enabled is False
    with HF.SpeculativeBackward():
        assert HF.SpeculativeBackward.enabled is True
        with HF.SpeculativeBackward():
            assert HF.SpeculativeBackward.enabled is True
        assert HF.SpeculativeBackward.enabled is True
    assert HF.SpeculativeBackward.enabled is False


def test_step_random_context_without_a_provider_changes_nothing():
    assert HF.StepRandom.active is None
    with HF.SpeculativeBackward(), HF.StepRandom.use(None) as prov:
        assert prov is None and HF.StepRandom.active is None
    assert HF.StepRandom.active is None


def test_step_random_context_resets_the_cursor_and_restores():
    class Fake:
        cursor = 7
    f = Fake()
    with HF.StepRandom.use(f) as prov:
        assert prov is f and HF.StepRandom.active is f and f.cursor == 0
    assert HF.StepRandom.active is None


def test_philox_checker_reproduces_the_random123_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output"""
    import numpy as np
    from . import helpers as H
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = H.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want
