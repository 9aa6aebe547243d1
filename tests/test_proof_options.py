"""ProofOptions presets and checked constructors (reference src/starks/proof/options.rs:35-151) through the C ABI, with the reference's own
unit tests (options.rs:164-270: F17 is too small a field; the generated 128 / 100 / 80-bit options pass their own check; one FRI query
less fails it) and the formulas recomputed in Python."""
import pytest

from lambdaworks_cairo_prover_amd import api

PO = api.ProofOptions
STARK252_BITS, F17_BITS = 252, 5         # F::field_bit_size()


def test_presets():
    want = {PO.CONJECTURABLE_80: 31, PO.CONJECTURABLE_100: 41, PO.CONJECTURABLE_128: 55, PO.PROVABLE_80: 80, PO.PROVABLE_100: 104, PO.PROVABLE_128: 140}
    for level, queries in want.items():
        o = PO.new_secure(level, 3)
        assert (o.blowup_factor, o.fri_number_of_queries, o.coset_offset, o.grinding_factor) == (4, queries, 3, 20)
    with pytest.raises(api.SpError):
        PO.new_secure(6, 3)
    assert PO.default_test_options() == PO(4, 3, 3, 1)


def test_u64_prime_field_is_not_large_enough_to_be_secure():          # options.rs:164-183
    o = PO.new_secure(PO.CONJECTURABLE_128, 1)
    with pytest.raises(api.SpError, match="InsecureOptionError::FieldSize"):
        PO.new_with_checked_security(o.blowup_factor, o.fri_number_of_queries, o.coset_offset, o.grinding_factor, 128, field_bits=F17_BITS)


@pytest.mark.parametrize("level,target", [(PO.CONJECTURABLE_128, 128), (PO.CONJECTURABLE_100, 100), (PO.CONJECTURABLE_80, 80)])
def test_generated_options_are_secure_for_their_target(level, target):   # options.rs:185-203, 228-270
    o = PO.new_secure(level, 1)
    assert PO.new_with_checked_security(o.blowup_factor, o.fri_number_of_queries, o.coset_offset, o.grinding_factor, target) == o


def test_one_fri_query_less_is_insecure():                              # options.rs:205-226
    o = PO.new_secure(PO.CONJECTURABLE_128, 1)
    with pytest.raises(api.SpError, match="InsecureOptionError::SecurityBits"):
        PO.new_with_checked_security(o.blowup_factor, o.fri_number_of_queries - 1, o.coset_offset, o.grinding_factor, 128)


def test_formulas_against_python():
    """options.rs:90-96 and :119-123 over a grid, the provable variant with the u8's leading zeros exactly as the reference has it."""
    for blowup in (2, 4, 8, 16, 64, 128):
        for queries in (1, 3, 31, 55, 80, 140):
            for grinding in (0, 1, 20):
                for target in (40, 80, 100, 128, 200):
                    field_ok = STARK252_BITS > target + 40
                    tz = (blowup & -blowup).bit_length() - 1
                    lz = 8 - blowup.bit_length()
                    for provable, insecure in ((False, target >= grinding + tz * queries - 1), (True, target < grinding + lz * queries // 2)):
                        try:
                            PO.new_with_checked_security(blowup, queries, 3, grinding, target, provable=provable)
                            ok = True
                        except api.SpError as e:
                            ok = False
                            assert ("FieldSize" in str(e)) == (not field_ok)
                        assert ok == (field_ok and not insecure), (blowup, queries, grinding, target, provable)
