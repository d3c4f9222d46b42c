"""Drop-in for the reference's ``ptsemseg.models`` boundary (models/__init__.py:8-101):
``get_model(cfg_dict, n_classes, version=None) -> nn.Module`` with the same YAML fields forwarded
as constructor keywords.  Only the architectures on the When2com hot path are provided
(SURVEY.md section 8a); the others raise with a pointer to why."""
from .srms import LearnWhen2Com, LearnWho2Com
from .when2com import MIMOcom, MIMOcomWho, Single_agent

_OUT_OF_SCOPE = ("All_agents", "MIMO_All_agents")


def _get_model_instance(name):
    table = {"Single_agent": Single_agent, "MIMOcom": MIMOcom, "MIMOcomWho": MIMOcomWho,
             "LearnWhen2Com": LearnWhen2Com, "LearnWho2Com": LearnWho2Com}
    if name in table:
        return table[name]
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError("arch %r is outside the accelerated When2com forward path (SURVEY.md section 8f)" % name)
    # the reference does `raise ("Model {} not available")` which surfaces as a TypeError
    raise TypeError("Model {} not available".format(name))


def get_model(model_dict, n_classes, version=None):
    name = model_dict["model"]["arch"]
    cls = _get_model_instance(name)
    m = model_dict["model"]
    in_channels = 3
    if name == "Single_agent":
        return cls(n_classes=n_classes, in_channels=in_channels, enc_backbone=m["enc_backbone"],
                   dec_backbone=m["dec_backbone"], feat_squeezer=m["feat_squeezer"], feat_channel=m["feat_channel"])
    if name in ("LearnWhen2Com", "LearnWho2Com"):            # models/__init__.py:36-57: agent_num arrives as aux_agent_num
        return cls(n_classes=n_classes, in_channels=in_channels, attention=m["attention"], has_query=m["query"],
                   sparse=m["sparse"], aux_agent_num=m["agent_num"], shared_img_encoder=m["shared_img_encoder"],
                   image_size=model_dict["data"]["img_rows"], query_size=m["query_size"], key_size=m["key_size"],
                   enc_backbone=m["enc_backbone"], dec_backbone=m["dec_backbone"])
    # MIMOcom / MIMOcomWho: only img_rows (not img_cols) reaches the model (models/__init__.py:64)
    return cls(n_classes=n_classes, in_channels=in_channels, attention=m["attention"], has_query=m["query"],
               sparse=m["sparse"], agent_num=m["agent_num"], shared_img_encoder=m["shared_img_encoder"],
               image_size=model_dict["data"]["img_rows"], query_size=m["query_size"], key_size=m["key_size"],
               enc_backbone=m["enc_backbone"], dec_backbone=m["dec_backbone"])
