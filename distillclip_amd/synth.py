"""Deterministic synthetic weights and inputs (counter-based, numpy only).

The same generator runs in the golden-generation container (where the reference
is importable) and on the GPU box (where it is not), so fixtures only need to
carry seeds + expected outputs.  Independent of torch's RNG on purpose
(SURVEY.md §8c/§8d).

Inputs follow SURVEY.md §8d: images ~ N(0,1) fp32 [B,3,R,R]; captions in
`clip.tokenize` layout (reference data/component/ms_coco.py:37): SOT 49406,
l~U{5..40} ids in 1..49405, EOT 49407 (= max id, so argmax picks it,
reference text_encoder.py:86), zero pad.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(seed, name):
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    h2 = zlib.adler32(name.encode()) & 0xFFFFFFFF
    return np.uint64(((h << 32) | h2) ^ ((seed * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF))


def _uniform_range(seed, name, lo, hi):
    with np.errstate(over='ignore'):
        idx = np.arange(lo, hi, dtype=np.uint64)
        z = _splitmix64(_splitmix64(idx ^ _key(seed, name)) + idx)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def uniform(seed, name, n):
    """n float64 uniforms in (0,1), a pure function of (seed, name, index)."""
    return _uniform_range(seed, name, 0, n)


_CHUNK = 1 << 18          # pairs per work item of a large tensor


def _workers():
    import os
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    return max(1, min(16, cpus))


def normal(seed, name, shape, std=1.0, mean=0.0):
    """Box-Muller on pairs (u1[i], u2[i]): element i < m is r cos, element m + i is r sin (m = ceil(n / 2)).  Every element is a pure function
    of (seed, name, index), so large tensors are produced in index chunks on a few threads (numpy releases the GIL) — same bits, the
    49408-row token tables of the real-shape tests in ~1 s instead of ~5."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    out = np.empty(n, dtype=np.float32)

    def piece(lo, hi):
        u1 = _uniform_range(seed, name + '#a', lo, hi)
        u2 = _uniform_range(seed, name + '#b', lo, hi)
        r = np.sqrt(-2.0 * np.log(u1))
        out[lo:hi] = (r * np.cos(2 * np.pi * u2) * std + mean).astype(np.float32)
        top = min(hi, n - m)
        if top > lo:
            out[m + lo:m + top] = (r[:top - lo] * np.sin(2 * np.pi * u2[:top - lo]) * std + mean).astype(np.float32)

    if m <= _CHUNK:
        piece(0, m)
    else:
        # short-lived threads, joined before returning (no pool: nothing of this module is alive at interpreter exit)
        import threading
        starts = list(range(0, m, _CHUNK))
        nw = min(_workers(), len(starts))
        errors = []

        def run(w):
            try:
                for lo in starts[w::nw]:
                    piece(lo, min(lo + _CHUNK, m))
            except BaseException as exc:      # re-raised on the calling thread
                errors.append(exc)
        threads = [threading.Thread(target=run, args=(w,), daemon=True) for w in range(nw)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    return out.reshape(shape)


def randint(seed, name, shape, lo, hi):
    """integers in [lo, hi] inclusive."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform(seed, name, n)
    return (lo + np.floor(u * (hi - lo + 1))).astype(np.int64).clip(lo, hi).reshape(shape)


# ---------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------
def images(seed, batch, resolution=224):
    return normal(seed, 'images', (batch, 3, resolution, resolution))


def captions(seed, batch, context_length=77, vocab_size=49408, min_len=5, max_len=40):
    sot, eot = vocab_size - 2, vocab_size - 1
    max_len = min(max_len, context_length - 2)
    min_len = min(min_len, max_len)
    lens = randint(seed, 'cap_len', (batch,), min_len, max_len)
    ids = randint(seed, 'cap_ids', (batch, context_length), 1, vocab_size - 3)
    out = np.zeros((batch, context_length), dtype=np.int64)
    for b in range(batch):
        l = int(lens[b])
        out[b, 0] = sot
        out[b, 1:1 + l] = ids[b, :l]
        out[b, 1 + l] = eot
    return out


# ---------------------------------------------------------------------------------------------
# weights, keyed by the reference's state_dict names
# ---------------------------------------------------------------------------------------------
def _ln(seed, sd, prefix, dim):
    sd[prefix + '.weight'] = normal(seed, prefix + '.weight', (dim,), 0.1, 1.0)
    sd[prefix + '.bias'] = normal(seed, prefix + '.bias', (dim,), 0.05)


def _teacher_blocks(seed, sd, prefix, width, layers):
    # stds: reference image_encoder.py:40-48 / text_encoder.py:98-106
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f'{prefix}transformer.resblocks.{i}.'
        sd[p + 'attn.in_proj_weight'] = normal(seed, p + 'attn.in_proj_weight', (3 * width, width), attn_std)
        sd[p + 'attn.in_proj_bias'] = normal(seed, p + 'attn.in_proj_bias', (3 * width,), attn_std)
        sd[p + 'attn.out_proj.weight'] = normal(seed, p + 'attn.out_proj.weight', (width, width), proj_std)
        sd[p + 'attn.out_proj.bias'] = normal(seed, p + 'attn.out_proj.bias', (width,), 0.02)
        _ln(seed, sd, p + 'ln_1', width)
        sd[p + 'mlp.c_fc.weight'] = normal(seed, p + 'mlp.c_fc.weight', (4 * width, width), fc_std)
        sd[p + 'mlp.c_fc.bias'] = normal(seed, p + 'mlp.c_fc.bias', (4 * width,), 0.02)
        sd[p + 'mlp.c_proj.weight'] = normal(seed, p + 'mlp.c_proj.weight', (width, 4 * width), proj_std)
        sd[p + 'mlp.c_proj.bias'] = normal(seed, p + 'mlp.c_proj.bias', (width,), 0.02)
        _ln(seed, sd, p + 'ln_2', width)


def teacher_image_state(seed, width=768, layers=12, patch=32, resolution=224, out_dim=512):
    """Keys of reference ImageEncoder (image_encoder.py:15: everything under `visual.`)."""
    sd = {}
    n_tok = (resolution // patch) ** 2 + 1
    sd['visual.conv1.weight'] = normal(seed, 'visual.conv1.weight', (width, 3, patch, patch), (3 * patch * patch) ** -0.5)
    sd['visual.class_embedding'] = normal(seed, 'visual.class_embedding', (width,), 0.02)
    sd['visual.positional_embedding'] = normal(seed, 'visual.positional_embedding', (n_tok, width), 0.01)
    _ln(seed, sd, 'visual.ln_pre', width)
    _teacher_blocks(seed, sd, 'visual.', width, layers)
    _ln(seed, sd, 'visual.ln_post', width)
    sd['visual.proj'] = normal(seed, 'visual.proj', (width, out_dim), width ** -0.5)
    return sd


def teacher_text_state(seed, width=512, layers=12, context_length=77, vocab_size=49408, out_dim=512):
    """Keys of reference TextEncoder (text_encoder.py:27-38)."""
    sd = {}
    sd['token_embedding.weight'] = normal(seed, 'token_embedding.weight', (vocab_size, width), 0.02)
    sd['positional_embedding'] = normal(seed, 'positional_embedding', (context_length, width), 0.01)
    _teacher_blocks(seed, sd, '', width, layers)
    _ln(seed, sd, 'ln_final', width)
    sd['text_projection'] = normal(seed, 'text_projection', (width, out_dim), width ** -0.5)
    return sd


def clip_student_states(seed, width, layers, patch, resolution, context_length, vocab_size, out_dim, tea_width_image, tea_width_text):
    """(image, text) state dicts of the reference's ImageEncoder(is_student=True) / TextEncoder(is_student=True): the CLIP tower's tensors
    plus the two projection linears nn.Linear(width, tea_transformer_width) (image_encoder.py:23-25, text_encoder.py:45-47)."""
    img = teacher_image_state(seed, width, layers, patch, resolution, out_dim)
    txt = teacher_text_state(seed, width, layers, context_length, vocab_size, out_dim)
    for tag, sd, tw in (('img', img, tea_width_image), ('txt', txt, tea_width_text)):
        for name in ('embedding_projection', 'hidden_projection'):
            sd[f'{name}.weight'] = normal(seed, f'{tag}.{name}.weight', (tw, width), width ** -0.5)
            sd[f'{name}.bias'] = normal(seed, f'{tag}.{name}.bias', (tw,), 0.02)
    return img, txt


def _student_blocks(seed, sd, prefix, dim, n_blocks, heads, repeats, mlp_ratio, qkv_bias, use_transform):
    hid = int(dim * mlp_ratio)
    eye = np.eye(heads, dtype=np.float32)
    for i in range(n_blocks):
        p = f'{prefix}blocks.{i}.block.'
        for r in range(repeats):
            _ln(seed, sd, p + f'norm1.instances.{r}', dim)
            _ln(seed, sd, p + f'norm2.instances.{r}', dim)
            if use_transform:
                # reference init is trunc_normal(.02) (weight_share_model.py:142-151); the synthetic
                # values are identity + noise so both head mixes are numerically exercised.
                for c in ('conv_l', 'conv_w'):
                    k = p + f'attn.{c}.instances.{r}.weight'
                    sd[k] = (eye + normal(seed, k, (heads, heads), 0.3 * heads ** -0.5)).reshape(heads, heads, 1, 1)
        sd[p + 'attn.qkv.weight'] = normal(seed, p + 'attn.qkv.weight', (3 * dim, dim), dim ** -0.5)
        if qkv_bias:
            sd[p + 'attn.qkv.bias'] = normal(seed, p + 'attn.qkv.bias', (3 * dim,), 0.02)
        sd[p + 'attn.proj.weight'] = normal(seed, p + 'attn.proj.weight', (dim, dim), dim ** -0.5)
        sd[p + 'attn.proj.bias'] = normal(seed, p + 'attn.proj.bias', (dim,), 0.02)
        sd[p + 'mlp.fc1.weight'] = normal(seed, p + 'mlp.fc1.weight', (hid, dim), dim ** -0.5)
        sd[p + 'mlp.fc1.bias'] = normal(seed, p + 'mlp.fc1.bias', (hid,), 0.02)
        sd[p + 'mlp.fc2.weight'] = normal(seed, p + 'mlp.fc2.weight', (dim, hid), hid ** -0.5)
        sd[p + 'mlp.fc2.bias'] = normal(seed, p + 'mlp.fc2.bias', (dim,), 0.02)


def student_image_state(seed, img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6,
                        num_heads=24, mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True, **_):
    """Keys of reference RepeatVisionTransformer (weight_share_model.py:226-315), repeated_times > 1."""
    assert repeated_times > 1, 'synthetic state covers the RepeatedMiniBlock key layout only'
    sd = {}
    n_tok = (img_size // patch_size) ** 2 + 1
    sd['cls_token'] = normal(seed, 's.cls_token', (1, 1, embed_dim), 0.02)
    sd['pos_embed'] = normal(seed, 's.pos_embed', (1, n_tok, embed_dim), 0.02)
    sd['patch_embed.proj.weight'] = normal(seed, 's.patch_embed.proj.weight',
                                           (embed_dim, in_chans, patch_size, patch_size),
                                           (in_chans * patch_size * patch_size) ** -0.5)
    sd['patch_embed.proj.bias'] = normal(seed, 's.patch_embed.proj.bias', (embed_dim,), 0.02)
    _student_blocks(seed, sd, '', embed_dim, depth // repeated_times, num_heads, repeated_times, mlp_ratio,
                    qkv_bias, use_transform)
    _ln(seed, sd, 'norm', embed_dim)
    sd['head.weight'] = normal(seed, 's.head.weight', (out_dim, embed_dim), embed_dim ** -0.5)
    sd['head.bias'] = normal(seed, 's.head.bias', (out_dim,), 0.02)
    return sd


def student_text_state(seed, vocab_size=49408, context_length=77, out_dim=512, embed_dim=768, depth=4,
                       num_heads=12, mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True,
                       compression_embedding=False, embedding_compression_dim=256, **_):
    """Keys of reference RepeatTextTransformer (weight_share_model.py:384-460), repeated_times > 1."""
    assert repeated_times > 1
    sd = {}
    if compression_embedding:
        sd['patch_embed.0.weight'] = normal(seed, 't.patch_embed.0.weight', (vocab_size, embedding_compression_dim), 0.5)
        sd['patch_embed.1.weight'] = normal(seed, 't.patch_embed.1.weight', (embed_dim, embedding_compression_dim),
                                            embedding_compression_dim ** -0.5)
        sd['patch_embed.1.bias'] = normal(seed, 't.patch_embed.1.bias', (embed_dim,), 0.02)
    else:
        sd['patch_embed.weight'] = normal(seed, 't.patch_embed.weight', (vocab_size, embed_dim), 0.5)
    sd['pos_embed'] = normal(seed, 't.pos_embed', (context_length, embed_dim), 0.02)
    _student_blocks(seed, sd, 't.', embed_dim, depth // repeated_times, num_heads, repeated_times, mlp_ratio,
                    qkv_bias, use_transform)
    # strip the disambiguating 't.' (only used to decorrelate the streams from the image student)
    sd = {(k[2:] if k.startswith('t.') else k): v for k, v in sd.items()}
    _ln(seed, sd, 'norm', embed_dim)
    sd['norm.weight'] = normal(seed, 't.norm.weight', (embed_dim,), 0.1, 1.0)
    sd['norm.bias'] = normal(seed, 't.norm.bias', (embed_dim,), 0.05)
    sd['head.weight'] = normal(seed, 't.head.weight', (out_dim, embed_dim), embed_dim ** -0.5)
    sd['head.bias'] = normal(seed, 't.head.bias', (out_dim,), 0.02)
    return sd
