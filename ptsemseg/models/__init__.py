from multiagentperception_amd.models import get_model, _get_model_instance  # noqa: F401
from multiagentperception_amd.models.when2com import MIMOcom, MIMOcomWho, Single_agent  # noqa: F401
from multiagentperception_amd.models.srms import LearnWhen2Com, LearnWho2Com  # noqa: F401
