"""State-dict layout of the COTR model (names + shapes), as one table.

The reference never writes this table down: it falls out of its module tree
(``COTR/models/cotr_model.py:17-24``, ``backbone.py:98-107`` + torchvision
``resnet50``, ``transformer.py:22-37,124-138,164-181``, ``position_encoding.py:14-21``).
Here it is explicit because three things need it: the parameter containers in
``cotr_model.py``, the weight packer behind ``cotr_load_weights`` and the synthetic
weight generator used by the benchmark and the parity tests.
"""
from collections import OrderedDict

# (planes, blocks, stride) of the ResNet-50 stages kept for each --layer value
_STAGES = [('layer1', 64, 3, 1), ('layer2', 128, 4, 2), ('layer3', 256, 6, 2), ('layer4', 512, 3, 2)]
LAYER_CHANNELS = {'layer1': 256, 'layer2': 512, 'layer3': 1024, 'layer4': 2048}
BN_FIELDS = ('weight', 'bias', 'running_mean', 'running_var')


def resnet_stages(layer):
    out = []
    for name, planes, blocks, stride in _STAGES:
        out.append((name, planes, blocks, stride))
        if name == layer:
            return out
    raise ValueError(f'unknown backbone layer {layer!r}')


def conv_bn_list(layer='layer3'):
    """[(conv_key, bn_key, cout, cin, k, stride)] in execution order, keys relative to
    ``backbone.0.body.``; block structure is recoverable from the key names."""
    out = [('conv1', 'bn1', 64, 3, 7, 2)]
    inplanes = 64
    for name, planes, blocks, stride in resnet_stages(layer):
        for b in range(blocks):
            s = stride if b == 0 else 1
            p = f'{name}.{b}.'
            out.append((p + 'conv1', p + 'bn1', planes, inplanes, 1, 1))
            out.append((p + 'conv2', p + 'bn2', planes, planes, 3, s))
            out.append((p + 'conv3', p + 'bn3', planes * 4, planes, 1, 1))
            if b == 0:
                out.append((p + 'downsample.0', p + 'downsample.1', planes * 4, inplanes, 1, s))
            inplanes = planes * 4
    return out


def state_spec(hidden_dim=256, nheads=8, enc_layers=6, dec_layers=6, dim_feedforward=1024,
               layer='layer3'):
    """OrderedDict name -> (shape, kind).  kind in {'conv','bn_w','bn_b','bn_rm','bn_rv',
    'bn_w_last','mat','bias','ln_w','ln_b','mlp_w','mlp_b'} (used only by the synthetic generator)."""
    d, f = hidden_dim, dim_feedforward
    spec = OrderedDict()

    def attn(prefix):
        spec[prefix + 'in_proj_weight'] = ((3 * d, d), 'mat')
        spec[prefix + 'in_proj_bias'] = ((3 * d,), 'bias')
        spec[prefix + 'out_proj.weight'] = ((d, d), 'mat')
        spec[prefix + 'out_proj.bias'] = ((d,), 'bias')

    def ffn_norms(prefix, norms):
        spec[prefix + 'linear1.weight'] = ((f, d), 'mat')
        spec[prefix + 'linear1.bias'] = ((f,), 'bias')
        spec[prefix + 'linear2.weight'] = ((d, f), 'mat')
        spec[prefix + 'linear2.bias'] = ((d,), 'bias')
        for n in norms:
            spec[prefix + n + '.weight'] = ((d,), 'ln_w')
            spec[prefix + n + '.bias'] = ((d,), 'ln_b')

    for i in range(enc_layers):
        p = f'transformer.encoder.layers.{i}.'
        attn(p + 'self_attn.')
        ffn_norms(p, ('norm1', 'norm2'))
    for i in range(dec_layers):
        p = f'transformer.decoder.layers.{i}.'
        attn(p + 'multihead_attn.')
        ffn_norms(p, ('norm1', 'norm2', 'norm3'))  # norm1 exists but is never applied (transformer.py:173,185-201)
    spec['transformer.decoder.norm.weight'] = ((d,), 'ln_w')
    spec['transformer.decoder.norm.bias'] = ((d,), 'ln_b')
    for i, (o, k) in enumerate([(d, d), (d, d), (2, d)]):
        spec[f'corr_embed.layers.{i}.weight'] = ((o, k), 'mlp_w')
        spec[f'corr_embed.layers.{i}.bias'] = ((o,), 'mlp_b')
    spec['input_proj.weight'] = ((d, LAYER_CHANNELS[layer], 1, 1), 'mat')
    spec['input_proj.bias'] = ((d,), 'bias')
    for conv, bn, cout, cin, k, _s in conv_bn_list(layer):
        spec[f'backbone.0.body.{conv}.weight'] = ((cout, cin, k, k), 'conv')
        for fld, kind in zip(BN_FIELDS, ('bn_w', 'bn_b', 'bn_rm', 'bn_rv')):
            if kind == 'bn_w' and bn.endswith('bn3'):
                kind = 'bn_w_last'  # last norm of a bottleneck (synthetic generator damps it)
            spec[f'backbone.0.body.{bn}.{fld}'] = ((cout,), kind)
    return spec
