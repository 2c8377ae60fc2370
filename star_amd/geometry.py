"""Shape helpers of the pipeline boundary (restating video_to_video/video_to_video_model.py:164-210)."""


def _centre_pad(size, target):
    a = int((target - size) // 2)
    return a, target - a - size


def pad_to_fit(h, w):
    """(w1, w2, h1, h2) constant padding that makes the frame a legal UNet input: small frames are centred on
    720x1280; larger ones grow to the next size with (H + 48) % 64 == 0 and W % 64 == 0, i.e. latent H = 2 (mod 8),
    W = 0 (mod 8) -- the sizes the Downsample pad (2,1) / Upsample row-crop pair round-trips (unet_v2v.py:564,709)."""
    best_h, best_w = 720, 1280
    if h < best_h:
        h1, h2 = _centre_pad(h, best_h)
    elif h == best_h:
        h1 = h2 = 0
    else:
        h1, h2 = 0, int((h + 48) // 64 * 64) + 64 - 48 - h
    if w < best_w:
        w1, w2 = _centre_pad(w, best_w)
    elif w == best_w:
        w1 = w2 = 0
    else:
        w1, w2 = 0, int(w // 64 * 64) + 64 - w
    return (w1, w2, h1, h2)


def sliding_windows_1d(length, window_size, overlap_size):
    """overlapping frame windows; the last window absorbs a tail shorter than 1/4 window."""
    stride = window_size - overlap_size
    ind, coords = 0, []
    while ind < length:
        if ind + window_size * 1.25 >= length:
            coords.append((ind, length))
            break
        coords.append((ind, ind + window_size))
        ind += stride
    return coords


def make_chunks(f_num, interp_f_num, max_chunk_len, chunk_overlap_ratio=0.5):
    max_o_len = max_chunk_len * chunk_overlap_ratio
    chunk_len = int((max_chunk_len - 1) // (1 + interp_f_num) * (interp_f_num + 1) + 1)
    o_len = int((max_o_len - 1) // (1 + interp_f_num) * (interp_f_num + 1) + 1)
    return sliding_windows_1d(f_num, chunk_len, o_len)
