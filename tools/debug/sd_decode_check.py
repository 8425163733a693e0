"""Host check of csrc/segdiam.hip's closed-form work-item numbering: every (row tile, strip) exactly once."""
import math
H, STRIP = 8, 16
for T in range(0, 300):
    q_, rem = divmod(T, STRIP)
    W = STRIP * q_ * (q_ + 1) // 2 + rem * (q_ + 1)
    seen = set()
    for q in range(W):
        g = int((math.sqrt(1.0 + 4.0 * q / H) - 1.0) * 0.5)
        while g > 0 and H * g * (g + 1) > q:
            g -= 1
        while H * (g + 1) * (g + 2) <= q:
            g += 1
        idx = q - H * g * (g + 1)
        r, strip = STRIP * g + 1 + idx // (g + 1), idx % (g + 1)
        ta = T - r
        assert 0 <= ta < T and ta + strip * STRIP < T, (T, q, ta, strip)
        seen.add((ta, strip))
    want = {(ta, st) for ta in range(T) for st in range(-(-(T - ta) // STRIP))}
    assert seen == want, T
print("ok")
