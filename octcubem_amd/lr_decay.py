"""Layer-wise learning-rate decay parameter groups for ViT fine-tuning (BEiT recipe).

Same contract as the reference's ``OCTCube/util/lr_decay.py``: ``param_groups_lrd(model, weight_decay, no_weight_decay_list,
layer_decay)`` returns ``[{"lr_scale", "weight_decay", "params"}, ...]`` for ``torch.optim``-style optimizers
(``lr_sched.adjust_learning_rate`` multiplies ``lr_scale`` in); ``get_layer_id_for_vit(name, num_layers)`` maps a parameter
name to its depth bucket: 0 = embeddings, i + 1 = ``blocks.i``, num_layers = everything after the blocks (norm, head)."""


def get_layer_id_for_vit(name, num_layers):
    if name in ("cls_token", "pos_embed") or name.startswith("patch_embed"):
        return 0
    if name.startswith("blocks"):
        return int(name.split(".")[1]) + 1
    return num_layers


def param_groups_lrd(model, weight_decay=0.05, no_weight_decay_list=(), layer_decay=0.75):
    num_layers = len(model.blocks) + 1
    layer_scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    groups = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        no_decay = p.ndim == 1 or name in no_weight_decay_list          # all 1-D parameters and the model's own list
        layer_id = get_layer_id_for_vit(name, num_layers)
        key = "layer_%d_%s" % (layer_id, "no_decay" if no_decay else "decay")
        if key not in groups:
            groups[key] = {"lr_scale": layer_scales[layer_id], "weight_decay": 0.0 if no_decay else weight_decay, "params": []}
        groups[key]["params"].append(p)
    return list(groups.values())
