"""LDS bank model (MI355X_MICROARCH.md, ds_read_b128: four 16-lane groups, 64 banks x 4 B) for k_cnn_trunk16's operand reads:
average LDS cycles per group relative to conflict-free, per layer, for candidate strides."""
groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def cost(addrs):
    tot = 0
    for g in groups:
        cnt = {}
        for l in g:
            cnt.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        tot += max(len(v) for v in cnt.values())
    return tot / 2.0


def conv1(RS):
    c = []
    for t in range(13):
        a = []
        for l in range(32):
            pos = min(32 * t + l, 399); oy, ox = divmod(pos, 20)
            a.append(4 * oy * RS + ox * 16)
        c.append(cost(a))
    return sum(c) / len(c)


def conv2(PIX):
    c = []
    for t in range(6):
        a = []
        for l in range(32):
            m = min(32 * t + l, 161); img, p = divmod(m, 81); oy, ox = divmod(p, 9)
            a.append(img * 400 * PIX + (4 * oy * 10 + ox) * PIX)
        c.append(cost(a))
    return sum(c) / len(c)


def conv3(PIX):
    c = []
    for t in range(4):
        a = []
        for l in range(32):
            m = min(32 * t + l, 97); img, p = divmod(m, 49); oy, ox = divmod(p, 7)
            a.append(img * 81 * PIX + (oy * 9 + ox) * PIX)
        c.append(cost(a))
    return sum(c) / len(c)


if __name__ == "__main__":
    for RS in (672, 688, 704, 720, 736, 752, 784): print("conv1 row stride", RS, round(conv1(RS), 2))
    for P in (64, 80, 96, 112, 144): print("conv2 pixel stride", P, round(conv2(P), 2))
    for P in (128, 144, 160, 176, 208): print("conv3 pixel stride", P, round(conv3(P), 2))


def conv2_rs(PIX, R0, IMG):
    c = []
    for t in range(6):
        a = []
        for l in range(32):
            m = min(32 * t + l, 161); img, p = divmod(m, 81); oy, ox = divmod(p, 9)
            a.append(img * IMG + 4 * oy * R0 + ox * PIX)
        c.append(cost(a))
    return sum(c) / len(c)


def conv3_rs(PIX, R1, IMG):
    c = []
    for t in range(4):
        a = []
        for l in range(32):
            m = min(32 * t + l, 97); img, p = divmod(m, 49); oy, ox = divmod(p, 7)
            a.append(img * IMG + oy * R1 + ox * PIX)
        c.append(cost(a))
    return sum(c) / len(c)


def search():
    best = []
    for PIX in (64, 80, 96, 112):
        for R0 in range(10 * PIX, 10 * PIX + 256, 16):
            for pad in range(0, 256, 16):
                IMG = 40 * R0 + pad
                best.append((round(conv2_rs(PIX, R0, IMG), 3), 2 * IMG, PIX, R0, IMG))
    best.sort()
    print("conv2 best (cost, bytes for 2 images, PIX, row stride, image stride):", best[:5])
    best = []
    for PIX in (128, 144, 160):
        for R1 in range(9 * PIX, 9 * PIX + 256, 16):
            for pad in range(0, 256, 16):
                IMG = 9 * R1 + pad
                best.append((round(conv3_rs(PIX, R1, IMG), 3), 2 * IMG, PIX, R1, IMG))
    best.sort()
    print("conv3 best:", best[:5])
