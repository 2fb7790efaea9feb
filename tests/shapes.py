"""state_dict key -> shape tables of the reference Generator / Discriminator
(SURVEY.md §8a row M*; model_probe_tune.py:373-457, 663-701), written out independently of
any module so that oracle, product and goldens can all be checked against one list."""
import math


def _ch(res, cm=2):
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm,
            512: 32 * cm, 1024: 16 * cm}[res]


def _styled(prefix, ci, co, sd, up):
    s = {f'{prefix}.conv.weight': (1, co, ci, 3, 3),
         f'{prefix}.conv.modulation.weight': (ci, sd),
         f'{prefix}.conv.modulation.bias': (ci,),
         f'{prefix}.noise.weight': (1,),
         f'{prefix}.activate.bias': (co,)}
    if up:
        s[f'{prefix}.conv.blur.kernel'] = (4, 4)
    return s


def _rgb(prefix, ci, sd, up):
    s = {f'{prefix}.bias': (1, 3, 1, 1),
         f'{prefix}.conv.weight': (1, 3, ci, 1, 1),
         f'{prefix}.conv.modulation.weight': (ci, sd),
         f'{prefix}.conv.modulation.bias': (ci,)}
    if up:
        s[f'{prefix}.upsample.kernel'] = (4, 4)
    return s


def generator_shapes(size, style_dim=512, n_mlp=8, cm=2):
    log = int(math.log2(size))
    s = {}
    for i in range(1, n_mlp + 1):
        s[f'style.{i}.weight'] = (style_dim, style_dim)
        s[f'style.{i}.bias'] = (style_dim,)
    s['input.input'] = (1, _ch(4, cm), 4, 4)
    s.update(_styled('conv1', _ch(4, cm), _ch(4, cm), style_dim, False))
    s.update(_rgb('to_rgb1', _ch(4, cm), style_dim, False))
    ci = _ch(4, cm)
    for j, i in enumerate(range(3, log + 1)):
        co = _ch(2 ** i, cm)
        s.update(_styled(f'convs.{2 * j}', ci, co, style_dim, True))
        s.update(_styled(f'convs.{2 * j + 1}', co, co, style_dim, False))
        s.update(_rgb(f'to_rgbs.{j}', co, style_dim, True))
        ci = co
    for l in range((log - 2) * 2 + 1):
        r = 2 ** ((l + 5) // 2)
        s[f'noises.noise_{l}'] = (1, 1, r, r)
    return s


def discriminator_shapes(size, cm=2):
    log = int(math.log2(size))
    s = {'convs.0.0.weight': (_ch(size, cm), 3, 1, 1), 'convs.0.1.bias': (_ch(size, cm),)}
    ci = _ch(size, cm)
    for b, i in enumerate(range(log, 2, -1), start=1):
        co = _ch(2 ** (i - 1), cm)
        s[f'convs.{b}.conv1.0.weight'] = (ci, ci, 3, 3)
        s[f'convs.{b}.conv1.1.bias'] = (ci,)
        s[f'convs.{b}.conv2.0.kernel'] = (4, 4)
        s[f'convs.{b}.conv2.1.weight'] = (co, ci, 3, 3)
        s[f'convs.{b}.conv2.2.bias'] = (co,)
        s[f'convs.{b}.skip.0.kernel'] = (4, 4)
        s[f'convs.{b}.skip.1.weight'] = (co, ci, 1, 1)
        ci = co
    s['final_conv.0.weight'] = (_ch(4, cm), ci + 1, 3, 3)
    s['final_conv.1.bias'] = (_ch(4, cm),)
    s['final_linear.0.weight'] = (_ch(4, cm), _ch(4, cm) * 16)
    s['final_linear.0.bias'] = (_ch(4, cm),)
    s['final_linear.1.weight'] = (1, _ch(4, cm))
    s['final_linear.1.bias'] = (1,)
    return s
