// String-keyed bag of untyped pointers: the argument convention of every operator and of
// OpticalFlow2D::ComputeFlow.  Interface of the reference's
// src/data_types/operation_parameters.h:28-38 (PushValuePtr does not overwrite an existing key).
#pragma once

#include <string>
#include <unordered_map>

class OperationParameters {
public:
    OperationParameters() = default;

    bool PushValuePtr(std::string key, void* value_ptr);
    void* GetValuePtr(std::string key) const;
    void Clear();

    // Typed read used by the operators: false (and `out` untouched) when the key is missing.
    template <typename T>
    bool Read(const char* key, T& out) const
    {
        void* p = GetValuePtr(key);
        if (!p) return false;
        out = *static_cast<T*>(p);
        return true;
    }

private:
    std::unordered_map<std::string, void*> map_;
};
