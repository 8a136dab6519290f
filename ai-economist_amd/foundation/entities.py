"""Game entities: resources, landmarks, endogenous quantities and agent classes.

Same names / flags as the reference (F/entities/resources.py:40-64,
F/entities/landmarks.py:56-88, F/entities/endogenous.py:32-36,
F/agents/mobiles.py, F/agents/planners.py); they only carry the metadata the
batched backend needs to lay out map channels and inventories.
"""
from .registrar import Registry


class Resource:
    name = ""
    color = None
    collectible = None


class Landmark:
    name = ""
    color = None
    ownable = None
    solid = True

    @property
    def blocking(self):
        return bool(self.solid and not self.ownable)

    @property
    def private(self):
        return bool(self.solid and self.ownable)

    @property
    def public(self):
        return bool(not self.solid and not self.ownable)


class Endogenous:
    name = ""


class Agent:
    name = ""


resource_registry = Registry(Resource)
landmark_registry = Registry(Landmark)
endogenous_registry = Registry(Endogenous)
agent_registry = Registry(Agent)


@resource_registry.add
class Wood(Resource):
    name = "Wood"
    collectible = True


@resource_registry.add
class Stone(Resource):
    name = "Stone"
    collectible = True


@resource_registry.add
class Coin(Resource):
    name = "Coin"
    collectible = False


for _res in ("Wood", "Stone"):

    @landmark_registry.add
    class _SourceBlock(Landmark):
        name = "{}SourceBlock".format(_res)
        ownable = False
        solid = False


@landmark_registry.add
class House(Landmark):
    name = "House"
    ownable = True
    solid = True


@landmark_registry.add
class Water(Landmark):
    name = "Water"
    ownable = False
    solid = True


@endogenous_registry.add
class Labor(Endogenous):
    name = "Labor"


@agent_registry.add
class BasicMobileAgent(Agent):
    name = "BasicMobileAgent"


@agent_registry.add
class BasicPlanner(Agent):
    name = "BasicPlanner"
