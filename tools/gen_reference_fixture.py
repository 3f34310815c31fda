"""Writes tests/golden/reference_value_cpp.json: the numbers the REFERENCE's own tests hold for this
path (they are data transcribed from the cited test files, not outputs of this repository):

* /root/reference/unittest/value.cpp:31-179 -- five RNEA vectors on models/simple_humanoid.urdf with a
  free-flyer root, gravity (0, 0, -9.8), cases 000 / 0V0 / 0VA / Q00 / QVA, with the tolerance each
  BOOST_CHECK uses (Eigen isApprox).  q = 0 there includes a ZERO quaternion, which the un-normalised
  quaternion formula maps to R = I (joint-free-flyer.hpp:338-339).
* /root/reference/unittest/joint-prismatic.cpp:137-180 (test_rnea) and :182-216 (test_crba): a single PX joint
  at the universe, Inertia(1, (0.5, 0, 0), I3), default gravity: tau = 0 / 1 / 1 and M = 1.

Inputs are stored explicitly next to the recipe they were built from.
"""
import json
import os

NQ, NV = 36, 35


def ramp_q(scale):  # for (i = 7; i < nq; ++i) q[i] = (i - 1) / scale
    return [0.0] * 7 + [(i - 1) / scale for i in range(7, NQ)]


def ramp_v(scale):  # for (i = 6; i < nv; ++i) v[i] = i / scale
    return [0.0] * 6 + [i / scale for i in range(6, NV)]


Z_Q, Z_V = [0.0] * NQ, [0.0] * NV

humanoid = [
    dict(name="000", cite="unittest/value.cpp:31-55", tol=1e-12, q=Z_Q, v=Z_V, a=Z_V,
         recipe="q = 0 (zero quaternion), v = 0, a = 0",
         expected=[0, 0, 1281.84, 0, -40.5132, 0, 4.4492, -1.5386, 0, -1.5386, -1.5386, 0, -4.4492,
                   -1.5386, 0, -1.5386, -1.5386, 0, -37.436, 0, 0, -2.548, 0, 0, 0.392, 0, 0.392, 0, -2.548, 0, 0,
                   0.392, 0, 0.392, 0]),
    dict(name="0V0", cite="unittest/value.cpp:57-85", tol=1e-7, q=Z_Q, v=ramp_v(10.0), a=Z_V,
         recipe="q = 0, v[i] = i/10 for i >= 6, a = 0",
         expected=[-48.10636, -73.218816, 1384.901025, -7.2292939, -186.342371, -20.6685852, -0.78887946,
                   -1.869651075, 1.8889752, -2.036768175, -2.105948175, -1.023232, -18.3738505, -4.133954895,
                   9.0456456, -4.276615035, -4.143427035, -2.534896, -180.338765, 15.71570676, -29.8639164,
                   -59.917862, -2.3307916, -2.7648728, -45.76782776, 3.4151272, -18.4320456, -4.5768072,
                   -76.60945104, -0.5897908, -2.640844, -63.93417064, 5.359156, -25.8309196, -6.976116]),
    dict(name="0VA", cite="unittest/value.cpp:87-115", tol=1e-6, q=Z_Q, v=ramp_v(100.0), a=ramp_v(10.0),
         recipe="q = 0, v[i] = i/100, a[i] = i/10 for i >= 6",
         expected=[-15.3331636, -0.67891816, 1273.80521, 102.9435113, 110.3509945, 81.52296995,
                   13.31476408, 14.26606068, 3.682505252, 9.048274318, 4.663303518, 2.05308568, 12.54347834,
                   25.92680911, 6.327105656, 16.71123385, 8.96650473, 3.54200704, 70.15812475, 77.02410963,
                   73.81994844, 41.73185754, 28.75786872, 28.94251127, 31.65847724, 20.40431127, 18.18579154,
                   6.838471928, 50.44193173, 34.07362801, 34.53507156, 38.33983417, 24.61507156, 22.2842788,
                   8.23435884]),
    dict(name="Q00", cite="unittest/value.cpp:117-148", tol=1e-6, q=ramp_q(10.0), v=Z_V, a=Z_V,
         recipe="q[i] = (i-1)/10 for i >= 7 (zero quaternion), v = 0, a = 0",
         expected=[5.367234435e-15, -2.587860481e-14, 1281.84, -133.3062501, 198.975587,
                   -7.120345979e-16, 15.06850407, 58.39287139, -22.14971864, 17.14327289, 1.291543104,
                   0.7402017048, -4.386231387, 22.73949408, -19.01681794, 0.8839600793, -0.3197599308,
                   -0.466827706, 65.47086697, 32.71449398, -4.250622066, -0.7937685568, -0.15349648, -1.070480752,
                   -3.066302263, -0.3557903212, -0.2183951073, 0.182684221, -0.6648425468, -2.902772493,
                   0.1250340934, 0.4402877138, -0.3158584741, -0.0865162794, 0.3918733239]),
    dict(name="QVA", cite="unittest/value.cpp:150-179", tol=1e-7, q=ramp_q(10.0), v=ramp_v(100.0), a=ramp_v(100.0),
         recipe="q[i] = (i-1)/10 for i >= 7, v[i] = a[i] = i/100 for i >= 6",
         expected=[-1.911650826, 1.250211406, 1284.82058, -139.0188156, 201.744449, 1.554847332,
                   15.1910084, 59.27339983, -21.70753738, 17.84339797, 1.754639468, 0.670280632, -2.968778954,
                   23.0776205, -17.56870284, 1.765761886, 0.2889992363, -0.392159764, 68.83598707, 34.59002827,
                   -4.604435817, -0.3832225891, 1.085231916, -0.348635267, -2.831371037, -1.047616506,
                   -0.228384161, 0.5656880079, 1.302869049, 0.8481280783, 0.7042182131, 1.554751317, -0.3908790552,
                   -0.1294643218, 1.421077555]),
]

prismatic = dict(
    cite="unittest/joint-prismatic.cpp:137-216",
    model=dict(joint="JointModelPX", parent=0, placement="identity", mass=1.0, lever=[0.5, 0.0, 0.0],
               inertia="identity", gravity=[0.0, 0.0, -9.81]),
    rnea=[dict(q=[0.0], v=[0.0], a=[0.0], expected=[0.0], tol=1e-14, absolute=True),
          dict(q=[1.0], v=[1.0], a=[1.0], expected=[1.0], tol=1e-12),
          dict(q=[3.0], v=[1.0], a=[1.0], expected=[1.0], tol=1e-12)],
    mass_matrix=[dict(q=[0.0], expected=[[1.0]], tol=1e-14),
                 dict(q=[1.0], expected=[[1.0]], tol=1e-12),
                 dict(q=[3.0], expected=[[1.0]], tol=1e-10)],
)

out = dict(
    source="Simple-Robotics/pinocchio unittest/value.cpp, unittest/joint-prismatic.cpp (numbers held by the reference's tests)",
    simple_humanoid=dict(urdf="tests/golden/simple_humanoid.urdf (copy of the reference's models/simple_humanoid.urdf)",
                         root_joint="JointModelFreeFlyer", gravity=[0.0, 0.0, -9.8], nq=NQ, nv=NV,
                         compare="Eigen isApprox: ||expected - tau|| <= tol * min(||expected||, ||tau||)",
                         cases=humanoid),
    prismatic_px=prismatic,
)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_value_cpp.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", os.path.normpath(path))
