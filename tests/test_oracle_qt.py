"""Oracle vs the reference's own quaternion unit tests (src/qt.rs:204-463).

Equality follows the reference: `==` on Quaternion is |a-b| < f64::EPSILON per component
(src/qt.rs:7-9,143-150); assert_eq! on f64 is exact.
"""
import math
import sys

import numpy as np

EPS = sys.float_info.epsilon


def qeq(a, b):
    return all(abs(x - y) < EPS for x, y in zip(a, b))


def test_sub_add_neg_scalar_conjugate(orc):
    q1, q2 = np.array([2.0, 0.0, 2.0, 0.0]), np.array([1.0, 0.0, 2.0, 1.0])
    assert list(q1 - q2) == [1.0, 0.0, 0.0, -1.0]                      # quaternion_sub
    assert list(np.array([2.0, -1.0, 2.0, 0.0]) + q2) == [3.0, -1.0, 4.0, 1.0]  # quaternion_add
    assert qeq(orc.q_conjugate([2.0, -1.0, 2.0, 0.0]), [2.0, 1.0, -2.0, 0.0])   # quaternion_conjugate


def test_eq_semantics():
    assert not qeq([1.0, 0, 0, 0], [1.000000000000001, 0, 0, 0])      # quaternion_eq
    assert qeq([1.0, 0, 0, 0], [1.0000000000000001, 0, 0, 0])


def test_mul(orc):
    q1, q2, q3 = [1.0, 0.0, 0.0, 2.0], [3.0, -1.0, 4.0, 3.0], [0.5, -3.0, 2.0, 9.0]
    assert qeq(orc.q_mul(q1, q2), [-3.0, -9.0, 2.0, 9.0])
    assert qeq(orc.q_mul(q2, q1), [-3.0, 7.0, 6.0, 9.0])
    assert qeq(orc.q_mul(orc.q_mul(q2, q1), q3), [-147.0 / 2.0, 97.0 / 2.0, -93.0, 19.0 / 2.0])


def test_conjugate_and_multiplication(orc):
    q1, q2 = [1.0, 0.0, 0.0, 2.0], [3.0, -1.0, 4.0, 3.0]
    assert qeq(orc.q_conjugate(orc.q_mul(q1, q2)), orc.q_mul(orc.q_conjugate(q2), orc.q_conjugate(q1)))
    assert qeq(orc.q_mul(orc.q_conjugate(q2), q2), [35.0, 0.0, 0.0, 0.0])


def test_dot_norm_normalize_inverse(orc):
    q = [math.sqrt(2.0) / 2.0, 0.0, math.sqrt(2.0) / 2.0, 0.0]
    assert orc.q_dot(q, q) == 1.0000000000000002                      # test_dot_product
    q1, q2 = [1.0, -3.0, 4.0, 3.0], [3.0, -1.0, 4.0, 3.0]
    assert orc.q_norm(q1) == 5.916079783099616                        # test_norm
    assert orc.q_norm(orc.q_mul(q1, q2)) == orc.q_norm(q1) * orc.q_norm(q2)
    assert qeq(orc.q_normalize(q1), [0.1690308509457033, -0.50709255283711, 0.6761234037828132, 0.50709255283711])
    inv = orc.q_inverse(orc.q_mul([1.0, 0.0, 0.0, 2.0], q2))
    assert qeq(inv, [-3.0 / 175.0, 9.0 / 175.0, -2.0 / 175.0, -9.0 / 175.0])


def test_distance(orc):
    q = [0.707106781, 0.0, 0.707106781, 0.0]
    assert orc.q_distance(q, q) == 0.0000000010552720919321246
    assert orc.q_distance(q, [0.707106781, 0.0, -0.707106781, 0.0]) == 1.0
    assert orc.q_distance(q, [0.0, 0.0, 1.0, 0.0]) == 0.5000000002638181
    assert orc.q_distance([1.0, 0.0, 0.0, 0.0], [0.5, 0.5, 0.5, 0.5]) == 0.75


def test_rotation(orc):
    v = orc.q_rotate([0.707106781, 0.0, 0.707106781, 0.0], [1.0, 0.0, 0.0])
    assert list(v) == [0.0, 0.0, -1.0]


def test_lerp(orc):
    q1, q2 = [1.0, 0.0, 0.0, 2.0], [3.0, -1.0, 4.0, 3.0]
    assert qeq(orc.q_lerp(q1, q2, 0.0), q1)
    assert qeq(orc.q_lerp(q1, q2, 1.0), q2)


def test_slerp(orc):
    q1, q2 = [1.0, 0.0, 0.0, 2.0], [3.0, -1.0, 4.0, 3.0]
    assert qeq(orc.q_slerp(q1, q2, 0.0), [0.4472135954999579, 0.0, 0.0, 0.8944271909999159])
    assert qeq(orc.q_slerp(q1, q2, 1.0), [0.50709255283711, -0.1690308509457033, 0.6761234037828132, 0.50709255283711])
    s = [0.7071067811865476, 0.0, 0.0, 0.7071067811865476]
    assert qeq(orc.q_slerp(s, s, 0.1), s)                                # test_slerp_same_quaternion
    assert qeq(orc.q_slerp([1.0, 0, 0, 0], [0, 0, 1.0, 0], 0.5), [0.7071067811865475, 0.0, 0.7071067811865475, 0.0])
    h = 0.7071067811865475
    assert qeq(orc.q_slerp([h, 0, 0, h], [0, h, h, 0], 0.5), [0.5, 0.5, 0.5, 0.5])


def test_random_quaternion_pins_stdrng(orc):
    """src/qt.rs:451-463: StdRng::seed_from_u64(324324324) -> fixed quaternion."""
    q = orc.Rng(324324324).quaternion()
    assert qeq(q, [0.31924330894562036, -0.5980633213833059, 0.5444724265858514, 0.49391674399349367])


def test_rng_stream_is_counter_addressable(orc):
    """u64 number k of the stream = words 2(k%8), +1 of ChaCha block k//8 (used by the GSO kernel)."""
    a, b = orc.Rng(324324), orc.Rng(324324)
    first = [a.next_u64() for _ in range(100)]
    assert len(set(first)) == 100
    assert [b.next_u64() for _ in range(100)] == first
    vals = [orc.Rng(7).f64() for _ in range(3)]
    assert vals[0] == vals[1] == vals[2] and 0.0 <= vals[0] < 1.0
