// ISerializable: the (de)serialisation half of the reference's Env interface (common/serializable.hpp:11-15).
#pragma once
#ifndef PPO_CPP_SERIALIZABLE_HPP        // (the reference's guard: see env/env.hpp)
#define PPO_CPP_SERIALIZABLE_HPP
#include "../json_min.hpp"

class ISerializable {
public:
    virtual ~ISerializable() {}
    virtual void serialize(nlohmann::json& json) = 0;
    virtual void deserialize(nlohmann::json& json) = 0;
};
#endif  // PPO_CPP_SERIALIZABLE_HPP
